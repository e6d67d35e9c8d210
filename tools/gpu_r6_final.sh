#!/bin/bash
# end-of-round measurements on ONE box: the bench line of every BASELINE configuration + the sharded path's one-GPU measurements.
# Results under gpurun_out/$1/ (copied into profiles/r06_*).
tag=${1:-r06final}
O=gpurun_out/$tag; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_cfg4_bf16.json 2>> $O/bench.err
for c in cfg1 cfg2 cfg3; do python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2>> $O/bench.err; done
python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_1gpu.json 2>> $O/bench.err
FASTEGNN_VIRT_CS_MAX_TILES=256 python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_1gpu_phased.json 2>> $O/bench.err
B="python bench.py --steps 50 --warmup 3 --no-cpu-baseline"
FASTEGNN_COMM=abi $B --sharded --hipgraph on > $O/sharded_w1_graph.json 2>> $O/bench.err
for w in 2 4 8; do FASTEGNN_COMM=abi $B --emulate-world $w --hipgraph on > $O/sharded_emu${w}_r0_sync.json 2>> $O/bench.err; done
FASTEGNN_COMM=abi python bench.py --config cfg5 --emulate-world 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/sharded_cfg5_emu8_r0.json 2>> $O/bench.err
python - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f"{f.split('/')[-1]:34s} ms/step {d['ms_per_step']:8.3f} eager {d.get('eager_ms_per_step')} value {d['value']} GB {d.get('peak_memory_gb')}")
    except Exception as e: print(f, "FAILED", e)
d = json.loads([l for l in open(sys.argv[1] + "/bench.json") if l.startswith("{")][0])
print(json.dumps(d["roofline"])); print(json.dumps(d["edge_scatter"])); print(d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu_baseline"))
PY
