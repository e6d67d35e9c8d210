#!/bin/bash
F="-mllvm -amdgpu-sched-strategy=max-memory-clause"
echo "== default build"; python tools/gpu_r5_sched_diag.py 2>&1 | tail -7
cd fastegnn_amd/csrc
rm -f layer_fwd.o; make layer_fwd.o EXTRA="$F" > /dev/null 2>&1; make -j16 ../libfastegnn_hip.so > /dev/null 2>&1
cd ../..
echo "== layer_fwd.hip with the flag"; python tools/gpu_r5_sched_diag.py 2>&1 | tail -7
