#!/bin/bash
# diagnostic: run-time sweep of the weight-gradient job geometry (FE_WG_CAP: most workgroups per job x batch, FE_WG_FILL)
for v in ${SWEEP:-"512 128" "256 128" "128 128" "64 128" "256 64" "256 32" "128 64"}; do
  set -- ${v/:/ }
  FE_WG_CAP=$1 FE_WG_FILL=$2 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('cap', $1, 'fill', $2, d['ms_per_step'], 'wgrad_tn', k['wgrad_tn_kernel']['ms_per_step'], 'reduce', k['wgrad_reduce_kernel']['ms_per_step'])"
done
