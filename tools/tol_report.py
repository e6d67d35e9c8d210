"""Reads a FASTEGNN_TOL_DUMP file (tests/helpers.py: grad_check) and prints every comparison that exceeds
GRAD_FACTOR x ref + GRAD_FLOOR, worst first: the input for tests/helpers.py GRAD_EXCEPTIONS.
    FASTEGNN_TOL_DUMP=gpurun_out/tol.jsonl python -m pytest tests -m gpu -q ; python tools/tol_report.py gpurun_out/tol.jsonl"""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
F, FL = 2.0, 1e-6
bad = [r for r in rows if r["got"] > F * r["ref"] + FL]
print(f"{len(rows)} comparisons, {len(bad)} beyond {F} x ref + {FL:g}")
for r in sorted(bad, key=lambda r: -(r["got"] - F * r["ref"])):
    print("  %-34s %-46s got %.2e ref %.2e excess %.2e  max|g| %.2e n=%d" % (
        r["case"][:34], r["tensor"][:46], r["got"], r["ref"], r["got"] - F * r["ref"], r["max"], r["numel"]))
