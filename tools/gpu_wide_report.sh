#!/bin/bash
# the wide path's price on one box: step times at hidden_nf = 128 (cfg4-shaped frames) and the rocprofv3 kernel table of the
# 20 000-node run.  Result: gpurun_out/wide_report/report.txt (copied to profiles/r04_wide_path.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide_report; mkdir -p $O
{
  echo "wide path (hidden_nf > 64; csrc/wide.hip + fastegnn_amd/wide.py), one MI355X, fwd + loss + bwd, eager launches"
  python $R/tools/gpu_wide_timing.py 20000 16 128 2>/dev/null | tail -1
  python $R/tools/gpu_wide_timing.py 100000 16 128 2>/dev/null | tail -1
  python $R/tools/gpu_wide_timing.py 100000 16 96 2>/dev/null | tail -1
  python $R/tools/gpu_wide_timing.py 20000 16 256 2>/dev/null | tail -1
} > $O/report.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_wide_timing.py 20000 16 128 > $O/log.txt 2>&1
cd $R
echo "" >> $O/report.txt
echo "rocprofv3 --kernel-trace --stats of: python3 tools/gpu_wide_timing.py 20000 16 128   (4 steps; Name, Calls, TotalDurationNs, AverageNs, Percentage)" >> $O/report.txt
head -16 $(ls $O/stats/*/*kernel_stats.csv | head -1) | cut -d, -f1-5 | cut -c1-200 >> $O/report.txt
cat $O/report.txt | cut -c1-220
