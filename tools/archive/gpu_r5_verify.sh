#!/bin/bash
# final-tree verification on ONE box: the whole -m gpu suite, smoke(), the default bench line, the wide path's report
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05verify; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/tests.txt; cat $O/tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
bash tools/gpu_r5_wide.sh > $O/wide.log 2>&1; head -8 $O/wide.log
