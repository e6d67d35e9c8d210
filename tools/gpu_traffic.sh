#!/bin/bash
# HBM traffic per kernel launch from PMC counters (separate passes, kernel-trace only), on ONE layer
# of the cfg4 frame (the profiler costs ~1 s per dispatch, so the run is kept to ~100 dispatches).
tag=${1:-traffic}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/$tag/$c -- python3 $R/bench.py --layers 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/$c.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/$tag/%s/*/*counter_collection.csv" % c)
    if not fs: print(c, "missing"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("fe::") and r["Counter_Name"] == c:
            k = k.replace("fe::", "").split("<")[0].replace("_pc_kernel", "_kernel").replace("_cs_kernel", "_kernel")   # profiler ids: one name per stage
            agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][c] = sum(v) / len(v)
        out[k]["launches"] = len(v)
res = {}
for k, v in out.items():
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads (x2)
    res[k] = {"fetch_KiB_raw": f, "write_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024, "launches_sampled": v.get("launches")}
json.dump(res, open("gpurun_out/$tag/traffic.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print(f"{k:28s} fetch(raw) {v['fetch_KiB_raw']/1024:9.1f} MiB  write {v['write_KiB']/1024:9.1f} MiB  -> HBM {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch")
PY
