#!/bin/bash
# repeat of the ring-slot A/B (each setting twice, interleaved) on ONE box
for rep in 1 2; do
for n in 4 3 2; do
  TAG="pcring${n}_$rep" EXTRA="-DFE_PC_RING=$n" BENCH_ARGS="" bash tools/gpu_variant_bench.sh | sed "s/^/rep $rep /"
  for r in 33 54; do
    FE_VB_RING=$r python bench.py --steps 40 --warmup 5 --cpu-baseline none 2>/dev/null | grep '{"metric"' > gpurun_out/var/tmp.json
    python - <<PY
import json
d = json.load(open("gpurun_out/var/tmp.json")); k = d["kernels"]
print("   rep $rep PC_RING=$n FE_VB_RING=$r ms/step", d["ms_per_step"], "virt_bwd", k["virt_bwd_kernel"]["ms_per_step"], "edge_bwd", k["edge_bwd_kernel"]["ms_per_step"])
PY
  done
done
done
cd fastegnn_amd/csrc && rm -f *.o && make -j8 ../libfastegnn_hip.so > /dev/null 2>&1
