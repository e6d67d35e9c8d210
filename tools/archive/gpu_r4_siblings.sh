#!/bin/bash
# FastRF / EGNN on the one remaining form of the virtual backward: their goldens + the FastEGNN parity / property tests,
# then the headline bench (the FastEGNN path must not move).
O=gpurun_out/sib; mkdir -p $O
python -m pytest tests/test_gpu_fastrf.py tests/test_gpu_egnn.py tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_bf16.py \
  tests/test_gpu_train.py -m gpu -q -x > $O/out.txt 2>&1
echo "exit $?" >> $O/out.txt
grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/out.txt | tail -25 | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/sib/bench.json").read().strip().splitlines()[-1])
print("cfg4 ms/step", d["ms_per_step"], "eager", d.get("eager_ms_per_step"))
PY
