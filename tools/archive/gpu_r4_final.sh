#!/bin/bash
# end-of-round captures on ONE box: bench lines of every configuration, rocprofv3 kernel stats of the default bench command, PMC
# traffic per launch, SQ counters, the gradient-tolerance report of the whole -m gpu suite.  Results under gpurun_out/$1/.
# (the sharded path's one-GPU measurements: tools/gpu_r4_sharded_final.sh)
tag=${1:-r04}
bash tools/gpu_round_profiles.sh $tag
bash tools/gpu_sq.sh $tag/sq > /dev/null 2>&1
rm -f gpurun_out/$tag/tol.jsonl
FASTEGNN_TOL_DUMP=gpurun_out/$tag/tol.jsonl python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -1 > gpurun_out/$tag/tol_pytest.txt
python tools/tol_report.py gpurun_out/$tag/tol.jsonl > gpurun_out/$tag/gradient_tolerance_report.txt
cat gpurun_out/$tag/tol_pytest.txt; head -4 gpurun_out/$tag/gradient_tolerance_report.txt
