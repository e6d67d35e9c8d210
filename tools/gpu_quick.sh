#!/bin/bash
# quick GPU check: parity tests + bench kernel table
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|^E " | head -8 | tee gpurun_out/pytest_tail.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bq.json 2>gpurun_out/bq.err || tail -5 gpurun_out/bq.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bq.json"))
print("graphs/s", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"])
tot=0
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:11]:
    tot+=v["ms_per_step"]; print(f'{k:26s} {v["ms_per_step"]:8.3f} ms  {v.get("tflops","")} TF  {v.get("gbs","")} GB/s')
print("sum top", tot)
PY
