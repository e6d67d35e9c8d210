#!/bin/bash
# usage on the GPU box: bash tools/gpu_bench.sh <tag>
set -x
tag=${1:-r01}
mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -c 3000 gpurun_out/$tag/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/$tag/prof -name "*stats*" | head
