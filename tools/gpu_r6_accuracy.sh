#!/bin/bash
# round 6: gradient error of the edge stage's parameter gradients at ~1 M edges (52 000 nodes of the cfg4 shape) per build variant:
# rebuilds the default library with each flag set on the box, runs the default-path oracle test with FASTEGNN_TOL_DUMP and prints
# every comparison above 8e-6.  usage: bash tools/gpu_r6_accuracy.sh "<flags A>" "<flags B>" ...
mkdir -p gpurun_out/acc
for extra in "$@"; do
  ( cd fastegnn_amd/csrc && rm -f layer_bwd.o virt_bwd.o misc.o && make -j16 ../libfastegnn_hip.so EXTRA="$extra" > /dev/null 2>&1 ) || { echo "build failed: $extra"; continue; }
  rm -f gpurun_out/acc/dump.jsonl
  FASTEGNN_TOL_DUMP=gpurun_out/acc/dump.jsonl python -m pytest tests/test_gpu_virt_cs.py -m gpu -q -k "size_where_it_is_the_default" 2>&1 | tail -1
  python - "$extra" <<'PY'
import json, sys
rows = [json.loads(l) for l in open("gpurun_out/acc/dump.jsonl")]
big = sorted([r for r in rows if r["got"] > 8e-6], key=lambda r: -r["got"])
print(f"[{sys.argv[1] or 'default'}] {len(rows)} comparisons, {len(big)} above 8e-6:")
for r in big[:14]: print(f"   {r['tensor']:34s} got {r['got']:.2e}  ref {r['ref']:.2e}  max|g| {r['max']:.2e}")
PY
done
( cd fastegnn_amd/csrc && rm -f layer_bwd.o virt_bwd.o misc.o && make -j16 ../libfastegnn_hip.so > /dev/null 2>&1 )
