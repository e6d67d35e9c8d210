#!/bin/bash
# measured lever, round 4, ONE box: ring slots of the in-workgroup consumers once the f16x2 images have freed 18 KB (edge_bwd_pc: two
# images) / 28 KB (virt_bwd_pc: three) of LDS.  virt_bwd_pc: FE_VB_RING=<ring A><ring B> at run time; edge_bwd_pc: -DFE_PC_RING=n.
mkdir -p gpurun_out/rings
cd fastegnn_amd/csrc && rm -f *.o && make -j8 ../libfastegnn_hip.so > /dev/null 2>&1 && cd ../..
for r in 33 43 44 54 55 64; do
  FE_VB_RING=$r python bench.py --steps 40 --warmup 5 --cpu-baseline none 2>/dev/null | grep '{"metric"' > gpurun_out/rings/vb$r.json
  python - <<PY
import json
d = json.load(open("gpurun_out/rings/vb$r.json")); k = d["kernels"]
print("FE_VB_RING=$r ms/step", d["ms_per_step"], "virt_bwd", k["virt_bwd_kernel"]["ms_per_step"], "edge_bwd", k["edge_bwd_kernel"]["ms_per_step"])
PY
done
for n in 5 3; do
  TAG="pcring$n" EXTRA="-DFE_PC_RING=$n" bash tools/gpu_variant_bench.sh
done
cd fastegnn_amd/csrc && rm -f *.o && make -j8 ../libfastegnn_hip.so > /dev/null 2>&1
