#!/bin/bash
# the sharded path's one-GPU measurements at the final tree (world 1, emulated ranks of 2 / 4 / 8, cfg5 emulated rank of 8) + one
# plain run of the whole -m gpu suite.  Results under gpurun_out/$1/.
tag=${1:-r04b}
B="python bench.py --steps 50 --warmup 3 --no-cpu-baseline"
mkdir -p gpurun_out/$tag/sharded
S=gpurun_out/$tag/sharded
$B > $S/unsharded.json 2> $S/err.txt
FASTEGNN_COMM=abi $B --sharded > $S/w1_eager.json 2>> $S/err.txt
FASTEGNN_COMM=abi $B --sharded --hipgraph on > $S/w1_graph.json 2>> $S/err.txt
for w in 2 4 8; do
  FASTEGNN_COMM=abi $B --emulate-world $w --hipgraph on > $S/emu${w}_r0_sync.json 2>> $S/err.txt
done
FASTEGNN_COMM=abi FASTEGNN_SHARDED_SYNC=0 $B --emulate-world 8 --hipgraph on > $S/emu8_r0_async_split.json 2>> $S/err.txt
FASTEGNN_COMM=abi $B --emulate-world 8 --emulate-rank 3 --hipgraph on > $S/emu8_r3_sync.json 2>> $S/err.txt
FASTEGNN_COMM=abi $B --emulate-world 8 --hipgraph off > $S/emu8_r0_sync_eager.json 2>> $S/err.txt
FASTEGNN_COMM=abi python bench.py --config cfg5 --emulate-world 8 --steps 10 --warmup 2 --no-cpu-baseline > $S/cfg5_emu8_r0.json 2>> $S/err.txt
if [ -z "$SKIP_SUITE" ]; then python -m pytest tests -m gpu -q > gpurun_out/$tag/pytest_full.txt 2>&1; echo "exit $?" >> gpurun_out/$tag/pytest_full.txt; fi
python - <<PY
import json, glob
for f in sorted(glob.glob("$S/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0]); k = d["kernels"]
        print(f"{f.split('/')[-1]:28s} ms/step {d['ms_per_step']:8.3f} eager {d.get('eager_ms_per_step')} launches {sum(v['launches_per_step'] for v in k.values()):.0f} kernel-sum {sum(v['ms_per_step'] for v in k.values()):.3f}")
    except Exception as e: print(f, "FAILED", e)
PY
[ -z "$SKIP_SUITE" ] && grep -v "amdgpu.ids" gpurun_out/$tag/pytest_full.txt | tail -4 | cut -c1-300
