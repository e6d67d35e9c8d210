#!/bin/bash
# what the driver runs at round end, on one box: the full -m gpu suite, smoke(), the default bench line
O=gpurun_out/verify; mkdir -p $O
python -m pytest tests -m gpu -q > $O/suite.txt 2>&1; echo "exit $?" >> $O/suite.txt
grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/suite.txt | tail -6 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/verify/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "dtype")}, d["roofline"]["frac"], d["roofline"]["frac_algorithmic"], d["edge_scatter"]["frac_of_hbm_peak"], d["cpu_baseline"]["value"])
PY
