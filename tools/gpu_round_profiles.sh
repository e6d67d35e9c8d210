#!/bin/bash
# end-of-round measurements on ONE box: bench lines of every configuration, rocprofv3 kernel stats of the default bench
# command, PMC traffic per launch.  Results under gpurun_out/$1/ (copy what is to be judged into profiles/).
tag=${1:-round}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/$tag/bench_cfg4_bf16.json 2>> gpurun_out/$tag/bench.err
for c in cfg1 cfg2 cfg3; do python bench.py --config $c --no-cpu-baseline > gpurun_out/$tag/bench_$c.json 2>> gpurun_out/$tag/bench.err; done
python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_cfg5_1gpu.json 2>> gpurun_out/$tag/bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$tag/bench*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["ms_per_step"], d.get("eager_ms_per_step"), d["value"], d["unit"])
    except Exception as e: print(f, "FAILED", e)
PY
python - <<PY
import torch
print("peak memory check: see bench_cfg5 line; torch reports max_memory_allocated inside the process only")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --hipgraph off > $R/gpurun_out/$tag/stats.log 2>&1
cd $R
cp $(ls gpurun_out/$tag/stats/*/*kernel_stats.csv | head -1) gpurun_out/$tag/kernel_stats.csv 2>/dev/null
head -25 gpurun_out/$tag/kernel_stats.csv | cut -c1-160
bash tools/gpu_traffic.sh $tag/traffic
