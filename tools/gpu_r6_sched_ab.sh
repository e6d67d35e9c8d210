O=$PWD/gpurun_out/sweep; mkdir -p $O
run() { timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[$1] ms/step', d['ms_per_step'], ' '.join('%s %.3f' % (n.replace('_kernel',''), k[n]['ms_per_step']) for n in ('edge_fwd_kernel','virt_fwd_kernel','edge_bwd_kernel','virt_bwd_kernel')))"; }
for rep in 1 2; do
  ( cd fastegnn_amd/csrc && rm -f virt_bwd.o layer_bwd.o && make -j16 ../libfastegnn_hip.so > /dev/null 2>&1 ); run "all mmc"
  ( cd fastegnn_amd/csrc && rm -f virt_bwd.o && make -j16 virt_bwd.o SCHED= > /dev/null 2>&1 && make ../libfastegnn_hip.so > /dev/null 2>&1 ); run "virt_bwd default sched"
  ( cd fastegnn_amd/csrc && rm -f layer_bwd.o && make -j16 layer_bwd.o SCHED= > /dev/null 2>&1 && make ../libfastegnn_hip.so > /dev/null 2>&1 ); run "virt_bwd + layer_bwd default sched"
done
