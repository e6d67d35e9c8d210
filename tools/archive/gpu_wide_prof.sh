#!/bin/bash
# rocprofv3 kernel stats of the wide path (hidden_nf = 128) on a 20 000-node frame
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wideprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_wide_timing.py 20000 16 128 > $O/log.txt 2>&1
cd $R
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
tail -2 $O/log.txt; head -22 $O/kernel_stats.csv | cut -c1-200
