mkdir -p gpurun_out/widerep; rm -f gpurun_out/widerep/*.jsonl
for i in $(seq 1 12); do FASTEGNN_TOL_DUMP=gpurun_out/widerep/run$i.jsonl python -m pytest tests/test_gpu_wide.py -m gpu -q -k "test_wide_model_vs_oracle and 160" 2>&1 | tail -1 > /dev/null; done
python - <<'PY'
import json, glob, collections
rows = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/widerep/run*.jsonl")):
    for l in open(f):
        d = json.loads(l); rows[d["tensor"]].append((d["got"], d["tol"], d["ref"]))
worst = sorted(rows.items(), key=lambda kv: -max(g / t for g, t, r in kv[1]))[:8]
for k, v in worst:
    print(f"{k:40s} tol {v[0][1]:.2e} ref {v[0][2]:.2e} got " + " ".join(f"{g:.2e}" for g, t, r in v))
PY
