#!/bin/bash
for fl in 0 256 1024 1280; do
  FASTEGNN_DEBUG_FLAGS=$fl python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$fl.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/ab_$fl.json')); k=d['kernels']
print('flags', $fl, 'edge_fwd', k['edge_fwd_kernel']['ms_per_step'], 'virt_fwd', k['virt_fwd_kernel']['ms_per_step'], 'step', d['ms_per_step'])"
done
