#!/bin/bash
# round 4: full -m gpu suite at the current tree, then the sharded path on ONE GPU: world-1 step (C-ABI transport, eager and as
# one HIP graph) against the unsharded step, and one emulated rank of an 8-rank partition (bench.py --emulate-world 8)
tag=${1:-r4s}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -12 > gpurun_out/$tag/pytest.txt
cat gpurun_out/$tag/pytest.txt
B="python bench.py --steps 50 --warmup 3 --no-cpu-baseline"
$B > gpurun_out/$tag/unsharded.json 2> gpurun_out/$tag/err.txt
FASTEGNN_COMM=abi $B --sharded > gpurun_out/$tag/w1_eager.json 2>> gpurun_out/$tag/err.txt
FASTEGNN_COMM=abi $B --sharded --hipgraph on > gpurun_out/$tag/w1_graph.json 2>> gpurun_out/$tag/err.txt
for r in 0 3; do
  FASTEGNN_COMM=abi FASTEGNN_SHARDED_SYNC=1 $B --emulate-world 8 --emulate-rank $r --hipgraph on > gpurun_out/$tag/emu8_r${r}_sync.json 2>> gpurun_out/$tag/err.txt
  FASTEGNN_COMM=abi FASTEGNN_SHARDED_SYNC=0 $B --emulate-world 8 --emulate-rank $r --hipgraph on > gpurun_out/$tag/emu8_r${r}_async.json 2>> gpurun_out/$tag/err.txt
done
FASTEGNN_COMM=abi FASTEGNN_SHARDED_SYNC=0 $B --emulate-world 8 --hipgraph off > gpurun_out/$tag/emu8_r0_async_eager.json 2>> gpurun_out/$tag/err.txt
FASTEGNN_COMM=abi FASTEGNN_SHARDED_SYNC=0 $B --emulate-world 2 --hipgraph on > gpurun_out/$tag/emu2_r0_async.json 2>> gpurun_out/$tag/err.txt
FASTEGNN_COMM=abi FASTEGNN_SHARDED_SYNC=0 $B --emulate-world 4 --hipgraph on > gpurun_out/$tag/emu4_r0_async.json 2>> gpurun_out/$tag/err.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$tag/*.json")):
    try:
        d = json.load(open(f))
        k = d["kernels"]
        print(f"{f.split('/')[-1]:28s} ms/step {d['ms_per_step']:7.3f} eager {d.get('eager_ms_per_step')} launches {sum(v['launches_per_step'] for v in k.values()):.0f} kernel-sum {sum(v['ms_per_step'] for v in k.values()):.3f}", d.get("shard", {}).get("edge_stage_launch_ranges"))
    except Exception as e: print(f, "FAILED", e)
PY
grep -v "amdgpu.ids" gpurun_out/$tag/err.txt | tail -5
