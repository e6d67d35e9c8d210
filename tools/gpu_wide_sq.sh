#!/bin/bash
# SQ counters of the wide path's operators (tools/gpu_wide_ops.py: every operator at cfg4's edge count, 6 launches each), three
# --pmc passes with kernel trace only.  Output: gpurun_out/wide_sq/sq_counters.txt (-> profiles/r05_wide_sq_counters.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide_sq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/tools/gpu_wide_ops.py > $O/$n.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU
run p2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM
cd $R
python - <<'PY' | tee gpurun_out/wide_sq/sq_counters.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(list)
for n in ("p1", "p2"):
    for f in glob.glob("gpurun_out/wide_sq/%s/*/*counter_collection.csv" % n):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if not k.startswith("fe::wide"): continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, n)].add(r["Dispatch_Id"])
    for f in glob.glob("gpurun_out/wide_sq/%s/*/*kernel_trace.csv" % n):
        if n != "p1": continue
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if k.startswith("fe::wide"): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --pmc (two passes, kernel trace only) of: python3 tools/gpu_wide_ops.py   (1 919 172 rows x 128 columns per operator)")
print("# per-LAUNCH averages summed over the chip; WAVE_CYCLES / ACTIVE_INST_* / WAIT_INST_* in units of 4 cycles summed over waves")
for k in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
    a = agg[k]
    def per(c, n): return a.get(c, 0.0) / (len(cnt[(k, n)]) or 1)
    wc = per("SQ_WAVE_CYCLES", "p1") or 1
    d = dur.get(k, [0.0])
    print(f"\n{k}   ({len(cnt[(k, 'p1')])} launches, {sum(d) / len(d):.0f} us each under the profiler)")
    print(f"  waves {per('SQ_WAVES', 'p1'):.4g}  VALU insts {per('SQ_INSTS_VALU', 'p1'):.4g}  MFMA insts {per('SQ_INSTS_MFMA', 'p1'):.4g}  LDS insts {per('SQ_INSTS_LDS', 'p2'):.4g}  VMEM insts {per('SQ_INSTS_VMEM', 'p2'):.4g}")
    print(f"  vector-issue share of a wave   ACTIVE_INST_VALU / WAVE_CYCLES  {per('SQ_ACTIVE_INST_VALU', 'p2') / wc:6.3f}")
    print(f"  MFMA pipe busy   VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs per CU...)  raw {per('SQ_VALU_MFMA_BUSY_CYCLES', 'p2'):.4g} busy cycles, SQ_BUSY_CYCLES {per('SQ_BUSY_CYCLES', 'p1'):.4g}")
    print(f"  waiting (any)  WAIT_INST_ANY / WAVE_CYCLES  {per('SQ_WAIT_INST_ANY', 'p1') / wc:6.3f}    on LDS  {per('SQ_WAIT_INST_LDS', 'p2') / wc:6.3f}")
    print(f"  LDS active / wave cycles {per('SQ_ACTIVE_INST_LDS', 'p2') / wc:6.3f}   bank conflict cycles / LDS active {per('SQ_LDS_BANK_CONFLICT', 'p2') / max(per('SQ_ACTIVE_INST_LDS', 'p2'), 1):6.3f}   VMEM active / wave cycles {per('SQ_ACTIVE_INST_VMEM', 'p2') / wc:6.3f}")
PY
