#!/bin/bash
# round-5 iteration check: the parity-bearing test files (not the whole suite) + a short cfg4 bench with the kernel table
O=gpurun_out/${1:-r5q}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_egnn.py tests/test_gpu_fastrf.py tests/test_gpu_sharded.py tests/test_gpu_toolkit.py ${EXTRA_TESTS} -m gpu -q -x > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt
grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/tests.txt | tail -12 | cut -c1-400
python bench.py --steps 100 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - $O <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("graphs/s", d["value"], "ms/step", d["ms_per_step"], "eager", d["eager_ms_per_step"])
print("roofline", d["roofline"]); print("edge_scatter", d["edge_scatter"])
tot = 0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += v["ms_per_step"]; print(f'{k:26s} {v["ms_per_step"]:8.4f} ms  {v["avg_launch_ms"]*1e3:8.1f} us/launch')
print("sum", round(tot, 3))
PY
