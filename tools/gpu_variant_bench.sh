#!/bin/bash
# diagnostic: rebuild the library with extra -D flags ($EXTRA) on the GPU box's scratch copy and print the per-kernel table of one bench run
# usage: EXTRA="-DFE_NT_STORE" TAG=nt bash tools/gpu_variant_bench.sh
cd fastegnn_amd/csrc && rm -f *.o && make -j8 ../libfastegnn_hip.so EXTRA="$EXTRA" > /dev/null 2>&1 && cd ../.. || exit 1
mkdir -p gpurun_out/var
python bench.py --steps 40 --warmup 5 --cpu-baseline none ${BENCH_ARGS} 2>/dev/null | grep '{"metric"' > gpurun_out/var/${TAG:-v}.json
python - <<PY
import json
d = json.load(open("gpurun_out/var/${TAG:-v}.json"))
k = d["kernels"]
print("${TAG:-v}", "ms/step", d["ms_per_step"], " ".join(f"{n.replace('_kernel','')}={v['ms_per_step']:.3f}" for n, v in k.items() if v["ms_per_step"] > 0.3))
PY
