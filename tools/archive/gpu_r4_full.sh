#!/bin/bash
# full -m gpu suite (twice: run-to-run noise at the tolerance thresholds) + the default bench line; results in gpurun_out/$tag/
tag=${1:-r4full}
mkdir -p gpurun_out/$tag
for i in 1 2; do
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -30 | cut -c1-400 > gpurun_out/$tag/pytest$i.txt
done
python bench.py --no-cpu-baseline > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/$tag/bench.json") if l.startswith("{")][0])
print("ms/step", d["ms_per_step"], "eager", d["eager_ms_per_step"], d["roofline"], d["edge_scatter"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]: print(f"  {k:26s} {v['ms_per_step']:8.3f}")
PY
cat gpurun_out/$tag/pytest1.txt gpurun_out/$tag/pytest2.txt
