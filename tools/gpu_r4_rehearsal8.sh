#!/bin/bash
# 4- and 8-rank rehearsal of `bench.py --gpus N` on ONE GPU (gloo transport, ranks share the device), launched as the driver
# launches it: exercises the 8-way partition, halo plans and both legs of the multi-rank bench on the HIP backend.
O=gpurun_out/reh8; mkdir -p $O
export FASTEGNN_BENCH_BACKEND=gloo
for n in 4 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29550 + n)) \
    bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline > $O/r$n.json 2> $O/r$n.err
  echo "N=$n exit $?"
  python - <<PY
import json
try:
    d = json.loads(open("$O/r$n.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("n_gpus", "ms_per_step", "value", "scaling")}, d["shard"], d["table_exchange"]["rows_received_per_exchange"])
except Exception as e:
    print("no json:", e)
PY
  grep -i "error\|Traceback" -A5 $O/r$n.err | head -20
done
