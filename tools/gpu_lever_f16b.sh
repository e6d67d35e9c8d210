#!/bin/bash
# measured levers, round 4, ONE box: f16x2 products in the forward kernels (-DFE_FWD_F16=7), in the backward producers too
# (-DFE_BWD_F16=3: per-item scaled gradient operands), and log2(e) folded into edge_fwd's packed weights (-DFE_LOG2E_FOLD=1).
# Per variant: bench line, the parity gate (failures of tests/test_gpu_parity.py + test_gpu_properties.py) and the error report.
mkdir -p gpurun_out/f16b
for v in ${VARIANTS:-"base:" "fwd:-DFE_FWD_F16=7" "fwd_fold:-DFE_FWD_F16=7 -DFE_LOG2E_FOLD=1" "all:-DFE_FWD_F16=7 -DFE_BWD_F16=3" "all_fold:-DFE_FWD_F16=7 -DFE_BWD_F16=3 -DFE_LOG2E_FOLD=1"}; do
  t="${v%%:*}"
  TAG="$t" EXTRA="${v#*:}" bash tools/gpu_variant_bench.sh
  cp gpurun_out/var/$t.json gpurun_out/f16b/
  [ "$t" = "base" ] && [ -z "$BASE_TESTS" ] && continue
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 | tail -12
  rm -f gpurun_out/f16b/tol_$t.jsonl
  FASTEGNN_TOL_DUMP=gpurun_out/f16b/tol_$t.jsonl python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q > /dev/null 2>&1
  python tools/tol_report.py gpurun_out/f16b/tol_$t.jsonl | head -14
done
cd fastegnn_amd/csrc && rm -f *.o && make -j8 ../libfastegnn_hip.so > /dev/null 2>&1
