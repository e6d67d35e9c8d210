#!/bin/bash
# round 4, first call: sanity of the sizing patch + where the world-1 sharded step stands against the unsharded one
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py tests/test_gpu_comm.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r4a/pytest.txt
python bench.py --steps 50 --warmup 3 --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
python bench.py --steps 50 --warmup 3 --no-cpu-baseline --sharded > gpurun_out/r4a/bench_sharded_eager.json 2>> gpurun_out/r4a/bench.err
python bench.py --steps 50 --warmup 3 --no-cpu-baseline --sharded --hipgraph on > gpurun_out/r4a/bench_sharded_graph.json 2>> gpurun_out/r4a/bench.err
python - <<'PY'
import json
for f in ("bench","bench_sharded_eager","bench_sharded_graph"):
    try:
        d=json.load(open(f"gpurun_out/r4a/{f}.json"))
        print(f, d["ms_per_step"], d.get("eager_ms_per_step"), sum(v["launches_per_step"] for v in d["kernels"].values()))
        for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"]): print(f"   {k:26s} {v['ms_per_step']:8.3f} {v['launches_per_step']}")
    except Exception as e: print(f,"FAILED",e)
PY
cat gpurun_out/r4a/pytest.txt; tail -3 gpurun_out/r4a/bench.err
