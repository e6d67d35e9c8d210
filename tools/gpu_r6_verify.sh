#!/bin/bash
# round 6: the full -m gpu suite, smoke() and the default bench line on the FINAL tree, one box -> gpurun_out/r06verify/ (profiles/r06_gpu_suite_head.txt)
O=gpurun_out/r06verify; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err
{ echo "# pytest -m gpu on the round-6 final tree, one MI355X box:"; cat $O/gpu_suite.txt; echo "# __graft_entry__.smoke():"; cat $O/smoke.txt; echo "# python bench.py (default: cfg4, 1 GPU):"; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06verify/bench.json") if l.startswith("{")][0])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "eager_ms_per_step", "n_gpus", "steps", "dtype")})
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "matrix_pipe_util", d["roofline"].get("matrix_pipe_util"), "edge_scatter", d["edge_scatter"]["frac_of_hbm_peak"], "cpu_baseline", d["cpu_baseline"]["value"])
PY
} | tee $O/summary.txt
