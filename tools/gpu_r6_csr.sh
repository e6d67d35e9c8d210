#!/bin/bash
# The counting form of the CSR build (csr.hip: count_index) against the radix sorts: the index tests, then build_csr's time inside the step at
# cfg4, on a 1/8 shard of it and at cfg2 / cfg3 (dense: the radix form stays), each with FASTEGNN_CSR_SORT=radix beside the default.
O=$PWD/gpurun_out/r06csr; mkdir -p $O; rm -f $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_toolkit.py -m gpu -q -x -k "csr" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
line() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d["kernels"]["build_csr"]
print("  ms_per_step %.3f  eager %.3f  build_csr %.1f us" % (d["ms_per_step"], d.get("eager_ms_per_step") or 0, 1e3 * k["ms_per_step"]))
PY
}
for form in default radix default radix; do
  [ $form = radix ] && export FASTEGNN_CSR_SORT=radix || unset FASTEGNN_CSR_SORT
  echo "== $form" | tee -a $O/ab.txt
  for w in 8 4; do
    FASTEGNN_COMM=abi timeout 300 python bench.py --emulate-world $w --hipgraph on --steps 50 --warmup 3 --no-cpu-baseline > $O/emu${w}_$form.json 2> $O/emu${w}_$form.err; echo -n "emu$w (replayed):" | tee -a $O/ab.txt; line $O/emu${w}_$form.json | tee -a $O/ab.txt
  done
  timeout 300 python bench.py --config cfg1 --steps 200 --warmup 5 --no-cpu-baseline > $O/cfg1_$form.json 2> $O/cfg1_$form.err; echo -n "cfg1:" | tee -a $O/ab.txt; line $O/cfg1_$form.json | tee -a $O/ab.txt
  timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/cfg4_$form.json 2> $O/cfg4_$form.err; echo -n "cfg4:" | tee -a $O/ab.txt; line $O/cfg4_$form.json | tee -a $O/ab.txt
done
unset FASTEGNN_CSR_SORT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_graphs.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/tests.txt
