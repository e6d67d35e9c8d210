#!/bin/bash
# round 6 against round 5 on ONE box: the round-5 tree (git archive 45e3900 under build/r05tree, its default library built on the box)
# and this tree, the default bench line (cfg4, 40 steps) alternating, two repeats each.
( cd build/r05tree/fastegnn_amd/csrc && make -j48 ../libfastegnn_hip.so > /dev/null 2>&1 ) || { echo "round-5 build failed"; exit 1; }
line() { python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernels']
print('[$1] ms/step', d['ms_per_step'], 'eager', d['eager_ms_per_step'], 'graphs/s', d['value'], ' '.join('%s %.3f' % (n.replace('_kernel',''), k[n]['ms_per_step']) for n in ('edge_fwd_kernel','virt_fwd_kernel','edge_bwd_kernel','virt_bwd_kernel')))"; }
for rep in 1 2 3; do
  ( cd build/r05tree && python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null ) | line "round 5 (45e3900)"
  python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | line "round 6 (this tree)"
done
