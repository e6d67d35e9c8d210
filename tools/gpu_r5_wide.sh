#!/bin/bash
# round 5: the wide path on the bf16x3 GEMM kernels (csrc/wide_gemm.h) -- operator and model parity, then the step times with
# the fused activations on and off, then the rocprofv3 kernel table of the 100 000-node run.  gpurun_out/wide_r5/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide_r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
if ! grep -q "passed" $O/tests.txt || grep -q "failed" $O/tests.txt; then echo "TESTS FAILED"; fi
{
  timeout 600 python $R/tools/gpu_wide_timing.py 20000 16 128 2>&1 | tail -1
  timeout 600 python $R/tools/gpu_wide_timing.py 100000 16 128 2>&1 | tail -1
  FASTEGNN_WIDE_FUSE=0 timeout 600 python $R/tools/gpu_wide_timing.py 100000 16 128 2>&1 | tail -1 | sed 's/^/FUSE=0: /'
  timeout 600 python $R/tools/gpu_wide_timing.py 100000 16 96 2>&1 | tail -1
  timeout 600 python $R/tools/gpu_wide_timing.py 20000 16 256 2>&1 | tail -1
} > $O/report.txt
cat $O/report.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_wide_timing.py ${1:-100000} 16 128 > $O/log.txt 2>&1
cd $R
f=$(ls $O/stats/*/*kernel_stats.csv | head -1)
head -24 $f | cut -d, -f1-5 | cut -c1-160 | tee $O/kernel_stats_head.txt
