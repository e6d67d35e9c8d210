#!/bin/bash
tag=${1:-t}
bash tools/gpu_quick.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
f=glob.glob("gpurun_out/$tag/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].split('(')[0]
    if n.startswith('fe::'):
        g=(n, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
        d[g].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for g,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    print(g, len(v), 'avg us %.1f'%(sum(v)/len(v)), 'tot ms %.2f'%(sum(v)/1e3))
PY
