#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (eager launches) -> gpurun_out/kstats/kernel_stats.csv (profiles/r05_kernel_stats.csv)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --hipgraph off > $O/stats.log 2>&1
cd $R
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
grep '{"metric"' $O/stats.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print({n: k[n]['avg_launch_ms'] for n in ('edge_fwd_kernel','edge_bwd_kernel','virt_bwd_kernel','virt_fwd_kernel')})"
head -12 $O/kernel_stats.csv | cut -c1-150
