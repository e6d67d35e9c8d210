#!/bin/bash
mkdir -p gpurun_out/j3
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_egnn.py tests/test_gpu_comm.py tests/test_gpu_train.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -12 | cut -c1-300
FASTEGNN_COMM=abi python bench.py --config cfg5 --emulate-world 8 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/j3/cfg5_emu8.json 2> gpurun_out/j3/err.txt
python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/j3/cfg5_full.json 2>> gpurun_out/j3/err.txt
python - <<'PY'
import json
for f in ("cfg5_full", "cfg5_emu8"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/j3/{f}.json") if l.startswith("{")][0]); k = d["kernels"]
        print(f, d["ms_per_step"], "kernel-sum", round(sum(v["ms_per_step"] for v in k.values()), 2), d.get("shard", {}).get("edge_stage_launch_ranges"), d.get("peak_memory_gb"))
        print("   ", " ".join(f"{n.replace('_kernel','')}={v['ms_per_step']:.2f}" for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"])[:10]))
    except Exception as e: print(f, "FAILED", e)
PY
bash tools/gpu_lever_f16b.sh
