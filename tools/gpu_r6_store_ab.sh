#!/bin/bash
# round 6, VERDICT item 1: store-vs-recompute A/B on the cfg4 frame, one box.  The levers are run-time switches of fastegnn_amd
# (FASTEGNN_EDGE_STORE / FASTEGNN_VIRT_STORE: the forward kernels store pre-activations, the backward kernels read them instead of
# recomputing the products); a parity run of the goldens under both switches comes first.
O=gpurun_out/$1; mkdir -p $O
FASTEGNN_EDGE_STORE=1 FASTEGNN_VIRT_STORE=1 FASTEGNN_VIRT_CS_MIN_GRID=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | tee $O/parity_store.txt
run() {
  env $1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>$O/err.txt | grep '{"metric"' > $O/b.json || tail -3 $O/err.txt
  python - "$1" $O/b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
names = ("edge_fwd_kernel", "virt_fwd_kernel", "edge_bwd_kernel", "virt_bwd_kernel")
print(f'[{sys.argv[1] or "recompute (default)"}] ms/step {d["ms_per_step"]} (eager {d["eager_ms_per_step"]}) peak {d.get("peak_memory_gb", "?")} GB: ' + "  ".join(f'{n.replace("_kernel","")} {k[n]["ms_per_step"]:.3f}' for n in names if n in k))
PY
}
for rep in 1 2; do
  run "FASTEGNN_EDGE_STORE=0"
  run "FASTEGNN_EDGE_STORE=1"
  run "FASTEGNN_VIRT_STORE=1"
  run "FASTEGNN_EDGE_STORE=1 FASTEGNN_VIRT_STORE=1"
done
