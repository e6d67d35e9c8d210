"""Reads FASTEGNN_TOL_DUMP files (tests/helpers.py: grad_check logs every gradient comparison with the tolerance it was
held to) and prints the comparisons closest to their tolerance: got / tol >= the threshold in ANY of the files, one row per
(case, tensor) with the values of every file next to each other.  The rows are the candidates for a run-to-run failure
of `pytest -m gpu -x` (sums of fp32 atomics in arrival order move by a few 1e-7 between runs).
    python tools/tol_margin.py [--min 0.8] run1.jsonl [run2.jsonl ...]"""
import json, sys
args = sys.argv[1:]
thr = 0.8
if args and args[0] == "--min":
    thr = float(args[1]); args = args[2:]
seen = {}
for i, fn in enumerate(args):
    for line in open(fn):
        r = json.loads(line)
        # a (case, tensor) pair can occur several times in one run (parametrised tests under one case name), each with its own
        # reference error and tolerance: the ratio is taken per record, the row shows the record with the largest one
        row = seen.setdefault((r["case"], r["tensor"]), {"tol": r["tol"], "ref": r["ref"], "ratio": 0.0, "got": [[] for _ in args]})
        if r["tol"] > 0 and r["got"] / r["tol"] > row["ratio"]:
            row["ratio"], row["tol"], row["ref"] = r["got"] / r["tol"], r["tol"], r["ref"]
        row["got"][i].append(r["got"] / r["tol"] if r["tol"] > 0 else 0.0)
rows = []
for (case, tensor), r in seen.items():
    if r["ratio"] >= thr:
        rows.append((r["ratio"], case, tensor, r))
print(f"{len(seen)} (case, tensor) pairs in {len(args)} file(s); {len(rows)} with got/tol >= {thr}")
for ratio, case, tensor, r in sorted(rows, reverse=True):
    vals = " ".join("%.2f" % max(g) if g else "-" for g in r["got"])
    print("  %.2f  %-36s %-40s tol %.2e ref %.2e  got/tol per file: %s" % (ratio, case[:36], tensor[:40], r["tol"], r["ref"], vals))
