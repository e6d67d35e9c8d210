#!/bin/bash
# round 6: machine-scheduler / codegen option sweep over the three stage translation units (layer_fwd, layer_bwd, virt_bwd), one box.
# usage: [FILES="layer_bwd.o virt_bwd.o"] bash tools/gpu_r6_sched_sweep.sh "<SCHED flags A>" "<SCHED flags B>" ...   ("-" = no scheduler option at all)
FILES=${FILES:-"layer_fwd.o layer_bwd.o virt_bwd.o"}
O=$PWD/gpurun_out/sweep; mkdir -p $O
i=0
for sch in "$@"; do
  [ "$sch" = "-" ] && sch=""
  ( cd fastegnn_amd/csrc && rm -f $FILES && make -j16 $FILES "SCHED=$sch" > $O/build$i.log 2>&1 && make ../libfastegnn_hip.so > $O/build$i.log 2>&1 ) || { echo "build failed: $sch"; tail -2 $O/build$i.log; i=$((i+1)); continue; }
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>$O/v$i.err | grep '{"metric"' > $O/v$i.json || tail -3 $O/v$i.err
  python - "$sch" $O/v$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
names = ("edge_fwd_kernel", "virt_fwd_kernel", "edge_bwd_kernel", "virt_bwd_kernel", "node_pre_bwd_kernel")
print(f'[{sys.argv[1] or "no option"}] ms/step {d["ms_per_step"]}: ' + "  ".join(f'{n.replace("_kernel","")} {k[n]["ms_per_step"]:.3f}' for n in names if n in k))
PY
  i=$((i+1))
done
( cd fastegnn_amd/csrc && rm -f $FILES && make -j16 ../libfastegnn_hip.so > /dev/null 2>&1 )
