#!/bin/bash
# measured lever: the forward kernels' products as a 2-part fp16 split (3 products, -DFE_FWD_F16=mask:
# 1 edge_fwd, 2 virt_fwd, 4 node_pre_fwd) against the 3-part bf16 split (6 products): step / kernel times and the gradient-error
# report of the parity + property tests, every build on ONE box
mkdir -p gpurun_out/f16
for v in ${VARIANTS:-"base:" "f16e:-DFE_FWD_F16=1" "f16all:-DFE_FWD_F16=7"}; do
  TAG="${v%%:*}" EXTRA="${v#*:}" bash tools/gpu_variant_bench.sh
  [ "${v%%:*}" = "f16all" ] && python tools/gpu_bf3.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tee gpurun_out/f16/bf3.txt
  rm -f gpurun_out/f16/tol_${v%%:*}.jsonl
  FASTEGNN_TOL_DUMP=gpurun_out/f16/tol_${v%%:*}.jsonl python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_toolkit.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -1
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q 2>&1 | grep -E "^FAILED" | cut -c1-200 | head -20
  python tools/tol_report.py gpurun_out/f16/tol_${v%%:*}.jsonl | head -30
  cp gpurun_out/var/${v%%:*}.json gpurun_out/f16/
done
cd fastegnn_amd/csrc && rm -f *.o && make -j8 ../libfastegnn_hip.so > /dev/null 2>&1
