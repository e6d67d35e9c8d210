#!/bin/bash
# One pass of the whole GPU suite with every gradient comparison logged (FASTEGNN_TOL_DUMP: nothing fails on a gradient in this mode, the
# other assertions stay), then the comparisons within 20 % of their tolerance (tools/tol_margin.py).  A second pass of the files that
# hold those rows shows which of them move between runs.
O=$PWD/gpurun_out/r06margin; mkdir -p $O; rm -f $O/*.jsonl
FASTEGNN_TOL_DUMP=$O/run1.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2 > $O/summary.txt
FASTEGNN_TOL_DUMP=$O/run2.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_virt_cs.py tests/test_gpu_egnn.py tests/test_gpu_fastrf.py tests/test_gpu_train.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1 >> $O/summary.txt
FASTEGNN_TOL_DUMP=$O/run3.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_virt_cs.py tests/test_gpu_egnn.py tests/test_gpu_fastrf.py tests/test_gpu_train.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1 >> $O/summary.txt
python tools/tol_margin.py --min 0.7 $O/run1.jsonl $O/run2.jsonl $O/run3.jsonl | tee -a $O/summary.txt
