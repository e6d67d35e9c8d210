#!/bin/bash
# SQ counters (separate passes, kernel trace only) on ONE layer of the cfg4 frame; per-kernel averages
tag=${1:-sq}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$tag/$n -- python3 $R/bench.py --layers 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/$n.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU
run p2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM
cd $R
python - <<PY
import csv, glob, collections
for n in ("p1","p2"):
    fs=glob.glob("gpurun_out/$tag/%s/*/*counter_collection.csv"%n)
    if not fs: print(n,"no file"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").split("<")[0].replace("_pc_kernel","_kernel")
        if not k.startswith("fe::"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    print("==",n)
    for k in ("fe::edge_fwd_kernel","fe::virt_fwd_kernel","fe::edge_bwd_kernel","fe::virt_bwd_kernel","fe::wgrad_tn_kernel"):
        if k in agg: print(k, len(cnt[k]), {a:("%.4g"%(b/len(cnt[k]))) for a,b in agg[k].items()})
PY
