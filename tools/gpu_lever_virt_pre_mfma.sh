#!/bin/bash
# measured lever (round 4): virt_fwd forms pre = A + Bc + vr * w_vr as a rank-2 update on the matrix pipe (4 v_mfma_f32_16x16x4_f32)
# instead of 32 vector operations per (tile, channel).  A = default build, B = -DFE_VIRT_PRE_MFMA (as run: the lever was the default of the tree and -DFE_VIRT_PRE_VALU the alternative).
# Interleaved A B A B on ONE box; then the parity gate on B.
O=gpurun_out/virt_pre_mfma; mkdir -p $O
cp fastegnn_amd/libfastegnn_hip.so $O/default.so
run() {  # tag
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '{"metric"' > $O/$1.json
  python - <<PY
import json
d = json.load(open("$O/$1.json"))
k = d["kernels"]
print("$1", "ms/step", d["ms_per_step"], " ".join(f"{n.replace('_kernel','')}={v['ms_per_step']:.3f}" for n, v in k.items() if v["ms_per_step"] > 0.3),
      "edge_scatter_frac", d["edge_scatter"]["frac_of_hbm_peak"])
PY
}
for rep in 1 2; do
  cp gpurun_alt_valu.so fastegnn_amd/libfastegnn_hip.so; run valu_$rep
  cp $O/default.so fastegnn_amd/libfastegnn_hip.so; run mfma_$rep
done
cp gpurun_alt_valu.so fastegnn_amd/libfastegnn_hip.so; python bench.py --config cfg3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '{"metric"' > $O/cfg3_valu.json
cp $O/default.so fastegnn_amd/libfastegnn_hip.so; python bench.py --config cfg3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '{"metric"' > $O/cfg3_mfma.json
python - <<'PY'
import json
for t in ("valu", "mfma"):
    d = json.load(open(f"gpurun_out/virt_pre_mfma/cfg3_{t}.json")); k = d["kernels"]
    print("cfg3", t, d["ms_per_step"], "edge_fwd", k["edge_fwd_kernel"]["ms_per_step"], "edge_bwd", k["edge_bwd_kernel"]["ms_per_step"])
PY
rm -f $O/default.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_bf16.py tests/test_gpu_fastrf.py -m gpu -q > $O/tests.txt 2>&1
grep -E "^FAILED|passed|failed" $O/tests.txt | cut -c1-250 | tail -8
