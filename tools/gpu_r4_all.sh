#!/bin/bash
# end-of-round capture on ONE box (tag $1, default r04d): tools/gpu_r4_final.sh (bench lines, rocprofv3 stats, PMC traffic, SQ counters,
# the full -m gpu suite with the tolerance dump) + the sharded path's one-GPU measurements (without a second suite run)
tag=${1:-r04d}
bash tools/gpu_r4_final.sh $tag
SKIP_SUITE=1 bash tools/gpu_r4_sharded_final.sh $tag
