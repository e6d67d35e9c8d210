#!/bin/bash
# measured lever: one Newton step on the sigmoid's reciprocal in the backward recompute (-DFE_SIGMOID_NEWTON): step time and
# the gradient-error report of the parity + property tests, for both builds on ONE box
for v in ${VARIANTS:-"base:" "newton:-DFE_SIGMOID_NEWTON"}; do
  TAG="${v%%:*}" EXTRA="${v#*:}" bash tools/gpu_variant_bench.sh
  rm -f gpurun_out/tol_${v%%:*}.jsonl
  FASTEGNN_TOL_DUMP=gpurun_out/tol_${v%%:*}.jsonl python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -1
  python tools/tol_report.py gpurun_out/tol_${v%%:*}.jsonl | head -25
done
