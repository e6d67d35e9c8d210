#!/bin/bash
# round 5: the wide path's price on one box -- step times at cfg4-shaped frames (fused activations on and off), the per-operator
# table, and the rocprofv3 kernel table of the 100 000-node run.  Result: gpurun_out/wide_r5/report.txt (-> profiles/r05_wide_path.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide_r5; mkdir -p $O; rm -rf $O/stats
cd $R
{
  echo "wide path (hidden_nf > 64; csrc/wide.hip + csrc/wide_gemm.h + fastegnn_amd/wide.py), one MI355X, fwd + loss + bwd, eager launches"
  timeout 600 python $R/tools/gpu_wide_timing.py 20000 16 128 2>/dev/null | tail -1
  timeout 600 python $R/tools/gpu_wide_timing.py 100000 16 128 2>/dev/null | tail -1
  FASTEGNN_WIDE_FUSE=0 timeout 600 python $R/tools/gpu_wide_timing.py 100000 16 128 2>/dev/null | tail -1 | sed 's/^/FASTEGNN_WIDE_FUSE=0 (every activation, head and segment sum its own launch): /'
  timeout 600 python $R/tools/gpu_wide_timing.py 100000 16 96 2>/dev/null | tail -1
  timeout 600 python $R/tools/gpu_wide_timing.py 20000 16 256 2>/dev/null | tail -1
  echo ""
  echo "per-operator timing at cfg4's edge count (tools/gpu_wide_ops.py: 1 919 172 rows x 128 columns; TB/s = algorithmic bytes / time)"
  timeout 300 python $R/tools/gpu_wide_ops.py 2>/dev/null | tail -22
} > $O/report.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_wide_timing.py 100000 16 128 > $O/log.txt 2>&1
cd $R
f=$(ls -t $O/stats/*/*kernel_stats.csv | head -1)
{
  echo ""
  echo "rocprofv3 --kernel-trace --stats of: python3 tools/gpu_wide_timing.py 100000 16 128   (4 steps; Name, Calls, TotalDurationNs, AverageNs, Percentage)"
  head -26 $f | cut -d, -f1-5 | sed 's/([^"]*"/"/' | cut -c1-150
} >> $O/report.txt
cp $f $O/kernel_stats.csv
cat $O/report.txt | cut -c1-200
