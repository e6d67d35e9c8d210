#!/bin/bash
# Repeats one pytest selection N times with FASTEGNN_TOL_DUMP and prints, per tensor matching PAT, the error of every run
# (run-to-run spread of a gradient whose summation order depends on the ticket order).
# usage: SEL="tests/test_gpu_properties.py -k cfg5_shape" PAT="virtual.0.bias" N=5 bash tools/gpu_tolrepeat.sh
SEL=${SEL:-"tests/test_gpu_properties.py -k cfg5_shape"}
PAT=${PAT:-"virtual.0.bias"}
N=${N:-5}
mkdir -p gpurun_out/tolrep; rm -f gpurun_out/tolrep/*.jsonl
for i in $(seq 1 $N); do
  FASTEGNN_TOL_DUMP=gpurun_out/tolrep/run$i.jsonl python -m pytest $SEL -m gpu -q -x 2>&1 | tail -1
done
python - "$PAT" <<'PY'
import json, glob, sys, collections
pat = sys.argv[1]
rows = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/tolrep/run*.jsonl")):
    for l in open(f):
        d = json.loads(l)
        if pat in d["tensor"]:
            rows[(d["case"], d["tensor"])].append(d)
for (c, t), v in sorted(rows.items()):
    print(f"{c[:34]:34s} {t:44s} ref {v[0]['ref']:.2e} tol {v[0]['tol']:.2e} got " + " ".join(f"{d['got']:.2e}" for d in v))
PY
