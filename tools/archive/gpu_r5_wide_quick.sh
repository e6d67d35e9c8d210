#!/bin/bash
# quick loop for the wide path: operator parity, per-operator timing, one model step time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide_quick; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q ${WIDE_TESTS:--k "linear or rowwise or act"} 2>&1 | tail -8
timeout 300 python tools/gpu_wide_ops.py 2>&1 | tail -20 | tee $O/ops.txt
timeout 600 python tools/gpu_wide_timing.py 100000 16 128 2>&1 | tail -1 | tee $O/step.txt
