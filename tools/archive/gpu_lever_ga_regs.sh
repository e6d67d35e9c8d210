#!/bin/bash
# measured lever (round 4): virt_bwd_pc keeps the tile's g_A in 16 registers across the channels of a unit (-DVB_GA_REGS) instead of
# accumulating it through memory per channel: step time, virt_bwd time, parity gate and the kernel's HBM traffic, ONE box
for v in "base:" "gareg:-DVB_GA_REGS"; do
  t="${v%%:*}"
  TAG="$t" EXTRA="${v#*:}" bash tools/gpu_variant_bench.sh
  bash tools/gpu_traffic.sh lever_ga_$t 2>/dev/null | grep -E "virt_bwd_kernel|edge_bwd_kernel"
  [ "$t" = "gareg" ] && python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-200 | tail -5
done
cd fastegnn_amd/csrc && rm -f *.o && make -j8 ../libfastegnn_hip.so > /dev/null 2>&1
