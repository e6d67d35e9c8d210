#!/bin/bash
# SQ counters (separate passes, kernel trace only: gpurun refuses --pmc together with other trace domains) on ONE layer of the cfg4
# frame; per-kernel per-launch averages and the derived ratios the DESIGN.md claims rest on.  Output: gpurun_out/$tag/sq_counters.txt
# (copy it to profiles/).  The program goes directly after `--` (no shell / env hop under the profiler).
tag=${1:-sq}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$tag/$n -- python3 $R/bench.py --layers 1 --steps 1 --warmup 1 --no-cpu-baseline --hipgraph off > $R/gpurun_out/$tag/$n.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU
run p2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM
run p3 SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
cd $R
python - <<PY | tee gpurun_out/$tag/sq_counters.txt
import csv, glob, collections, subprocess, json, os
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for n in ("p1", "p2", "p3"):
    fs = glob.glob("gpurun_out/$tag/%s/*/*counter_collection.csv" % n)
    if not fs:
        print("#", n, "no counter file (see gpurun_out/$tag/%s.log)" % n); continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith("fe::"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, n)].add(r["Dispatch_Id"])
print("# rocprofv3 --pmc (three passes, kernel trace only) of: python3 bench.py --layers 1 --steps 1 --warmup 1 --hipgraph off")
print("# cfg4 frame (100 000 nodes, 1.92 M edges, C = 16), ONE layer, fp32 mode; values are per-LAUNCH averages summed over the chip.")
print("# commit:", subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "(snapshot without .git)")
OUT = {}
names = ["fe::edge_fwd_kernel", "fe::virt_fwd_kernel", "fe::edge_bwd_pc_kernel", "fe::virt_bwd_cs_kernel", "fe::virt_bwd_pc_kernel", "fe::virt_bwd_gv_kernel",
         "fe::virt_bwd_node_kernel", "fe::node_pre_fwd_kernel", "fe::node_pre_bwd_kernel", "fe::wgrad_tn_kernel", "fe::wgrad_reduce_kernel"]
for k in names:
    if k not in agg: continue
    a = agg[k]
    def per(c, n):
        l = len(cnt[(k, n)]) or 1
        return a.get(c, 0.0) / l
    v = {c: per(c, n) for n, cs in (("p1", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU"]),
                                    ("p2", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM"]),
                                    ("p3", ["SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE"])) for c in cs}
    print(f"\n{k}  ({len(cnt[(k, 'p1')])} launches sampled)")
    for c, x in v.items():
        print(f"  {c:30s} {x:16.4g}")
    # Units (checked on edge_fwd: 5.85 M MFMAs x 16 cycles = VALU_MFMA_BUSY_CYCLES exactly): *_BUSY_CYCLES in cycles; WAVE_CYCLES,
    # ACTIVE_INST_* and WAIT_INST_* in units of 4 cycles, summed over waves.  ACTIVE_INST_VALU / WAVE_CYCLES is the share of a wave's
    # resident time in which it issues vector / MFMA instructions; times the waves per SIMD it is the SIMD's vector-issue utilisation.
    wps = {"fe::edge_fwd_kernel": 4, "fe::virt_fwd_kernel": 2, "fe::edge_bwd_pc_kernel": 2, "fe::virt_bwd_pc_kernel": 2, "fe::virt_bwd_cs_kernel": 2,
           "fe::node_pre_fwd_kernel": 2, "fe::node_pre_bwd_kernel": 2}.get(k)
    wc = v["SQ_WAVE_CYCLES"] or 1
    share = v["SQ_ACTIVE_INST_VALU"] / wc
    print("  -- derived")
    print(f"  VALU instructions per MFMA instruction                   {v['SQ_INSTS_VALU'] / max(v['SQ_INSTS_MFMA'], 1):8.2f}")
    print(f"  issue model (4 N_valu + 8 N_mfma) / (4 ACTIVE_INST_VALU)    {(4 * v['SQ_INSTS_VALU'] + 8 * v['SQ_INSTS_MFMA']) / max(4 * v['SQ_ACTIVE_INST_VALU'], 1):8.3f}   (1.0: the issue port is what ACTIVE_INST_VALU counts)")
    print(f"  vector-issue share of a wave  ACTIVE_INST_VALU / WAVE_CYCLES {share:8.3f}" + (f"   x {wps} waves per SIMD = {share * wps:.2f} of the SIMD's issue slots" if wps else ""))
    if wps:
        print(f"  MFMA pipe busy  VALU_MFMA_BUSY_CYCLES / (4 WAVE_CYCLES / {wps})   {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc / wps):8.3f}   (of the SIMD's time)")
    print(f"  waiting (any)   WAIT_INST_ANY / WAVE_CYCLES                {v['SQ_WAIT_INST_ANY'] / wc:8.3f}")
    print(f"  waiting on LDS  WAIT_INST_LDS / WAVE_CYCLES                {v['SQ_WAIT_INST_LDS'] / wc:8.3f}")
    print(f"  LDS bank conflicts  LDS_BANK_CONFLICT / ACTIVE_INST_LDS    {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_ACTIVE_INST_LDS'], 1):8.3f}")
    J = dict(v)
    J["waves_per_simd"] = wps
    J["mfma_busy"] = (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc / wps)) if wps else None
    J["valu_issue_share_of_simd"] = share * wps if wps else None
    J["lds_bank_conflict_ratio"] = v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"], 1)
    J["launches_sampled"] = len(cnt[(k, "p1")])
    OUT[k] = J
OUT["_meta"] = {"commit": os.environ.get("GRAFT_COMMIT", ""), "what": "rocprofv3 --pmc, three passes, kernel trace only (tools/gpu_sq.sh): per-LAUNCH "
                "averages summed over the chip, ONE layer of the cfg4 frame, fp32 mode; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / the SIMDs' time; "
                "SQ_INSTS_VALU_MFMA_MOPS_* count 512 FLOP each"}
json.dump(OUT, open("gpurun_out/$tag/sq_counters.json", "w"), indent=1)
PY
