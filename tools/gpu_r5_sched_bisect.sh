#!/bin/bash
# which translation unit breaks under -mllvm -amdgpu-sched-strategy=max-memory-clause?  One file at a time with the flag, smoke() each.
F="-mllvm -amdgpu-sched-strategy=max-memory-clause"
cd fastegnn_amd/csrc
for f in ${@:-pack misc csr layer_fwd layer_bwd virt_bwd train}; do
  rm -f $f.o
  make $f.o EXTRA="$F" > /dev/null 2>&1 || echo "build failed"
  make -j16 ../libfastegnn_hip.so > /dev/null 2>&1
  echo "== $f with the flag: $(cd ../.. && python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | cut -c1-200)"
  rm -f $f.o; make $f.o > /dev/null 2>&1
done
