#!/bin/bash
# diagnostic: rebuild given sources with extra -D flags (on the GPU box's scratch copy) and print kernel times
# usage: tools/gpu_variants.sh "<files>" "<flags variant 1>" "<flags variant 2>" ...
files=$1; shift
for extra in "$@"; do
  ( cd fastegnn_amd/csrc && for f in $files; do rm -f $f.o; done && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $extra" > /dev/null 2>&1 ) || { echo "build failed: $extra"; continue; }
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || tail -5 gpurun_out/bv.err
  python - "$extra" <<'PY'
import json, sys
d=json.load(open("gpurun_out/bv.json"))
k=d["kernels"]
print(f'[{sys.argv[1]}] ms/step {d["ms_per_step"]}: ' + "  ".join(f'{n.replace("_kernel","")} {k[n]["ms_per_step"]:.3f}' for n in ("edge_fwd_kernel","virt_fwd_kernel","edge_bwd_kernel","virt_bwd_kernel","wgrad_tn_kernel","node_pre_fwd_kernel","node_pre_bwd_kernel") if n in k))
PY
done
