#!/bin/bash
# Full GPU suite at HEAD + the two-rank rehearsal of `bench.py --gpus 2` on one GPU (gloo transport, ranks share the device):
# both launch styles -- bench.py spawning its ranks, and the driver's `python -m torch.distributed.run`.
O=gpurun_out/suite_h; mkdir -p $O
python -m pytest tests -m gpu -q > $O/out.txt 2>&1
echo "exit $?" >> $O/out.txt
grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/out.txt | tail -15 | cut -c1-300
export FASTEGNN_BENCH_BACKEND=gloo
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/rehearsal_spawn.json 2> $O/rehearsal_spawn.err
echo "spawn exit $?"; tail -c 1500 $O/rehearsal_spawn.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/rehearsal_torchrun.json 2> $O/rehearsal_torchrun.err
echo "torchrun exit $?"; tail -c 600 $O/rehearsal_torchrun.json; tail -5 $O/rehearsal_torchrun.err | cut -c1-300
