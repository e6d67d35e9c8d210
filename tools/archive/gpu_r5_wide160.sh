R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide160; mkdir -p $O; rm -rf $O/stats
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_wide_timing.py 20000 16 160 > $O/log.txt 2>&1
tail -1 $O/log.txt
