#!/bin/bash
# round-5 A/B: rebuild the DEFAULT library with each variant's extra -D flags (on the GPU box's scratch copy) and print the kernel
# table of a 40-step cfg4 bench.  usage: bash tools/gpu_r5_ab.sh <tag> "<flags A>" "<flags B>" ...   ("" = the tree's defaults)
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
i=0
for extra in "$@"; do
  envv=""
  case "$extra" in ENV:*) envv="${extra#ENV:}"; extra="";; esac
  ( cd fastegnn_amd/csrc && ls *.o | grep -v "^wide" | xargs rm -f && make -j16 ../libfastegnn_hip.so EXTRA="$extra" > /dev/null 2>&1 ) || { echo "build failed: $extra"; continue; }
  env $envv timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>$O/v$i.err | grep '{"metric"' > $O/v$i.json || tail -3 $O/v$i.err
  python - "$extra$envv" $O/v$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernels"]
names = ("edge_fwd_kernel", "virt_fwd_kernel", "edge_bwd_kernel", "virt_bwd_kernel", "virt_bwd_gv_kernel", "wgrad_tn_kernel", "wgrad_reduce_kernel", "node_pre_bwd_kernel")
print(f'[{sys.argv[1] or "default"}] ms/step {d["ms_per_step"]} (eager {d["eager_ms_per_step"]}): ' + "  ".join(f'{n.replace("_kernel","")} {k[n]["ms_per_step"]:.3f}' for n in names if n in k))
PY
  i=$((i+1))
done
# leave the tree's default build behind
( cd fastegnn_amd/csrc && ls *.o | grep -v "^wide" | xargs rm -f && make -j16 ../libfastegnn_hip.so > /dev/null 2>&1 )
