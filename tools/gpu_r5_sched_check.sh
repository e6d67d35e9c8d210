#!/bin/bash
# The inline-assembly split blocks end with the VALU -> MFMA-operand wait states (common.h: mix_pack4).  Check: the tree's build, then
# the default library rebuilt with -mllvm -amdgpu-sched-strategy=max-memory-clause (the schedule that exposed the missing pad):
# smoke(), the small-graph diagnostic against the fp64 oracle, and a 40-step bench of each.
F="-mllvm -amdgpu-sched-strategy=max-memory-clause"
run() {
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
  python tools/gpu_r5_sched_diag.py 2>&1 | tail -7
  python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('ms/step', d['ms_per_step'], ' '.join(f'{n[:-7]} {k[n][\"ms_per_step\"]:.3f}' for n in ('edge_fwd_kernel','virt_fwd_kernel','edge_bwd_kernel','virt_bwd_kernel')))"
}
echo "== the tree's build"; run
cd fastegnn_amd/csrc; ls *.o | grep -v "^wide" | xargs rm -f; make -j16 ../libfastegnn_hip.so EXTRA="$F" > /dev/null 2>&1 || echo "build failed"; cd ../..
echo "== every stage kernel with $F"; run
