#!/bin/bash
# PMC passes (counters only + kernel trace), results summarised per kernel
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$tag/$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/$n.log 2>&1
}
mkdir -p $R/gpurun_out/$tag
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run p2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32
run p3 FETCH_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run p4 WRITE_SIZE TCC_EA0_ATOMIC_sum
cd $R
python - <<PY
import csv, glob, collections
for n in ("p1","p2","p3","p4"):
    fs=glob.glob("gpurun_out/$tag/%s/*/*counter_collection.csv"%n)
    if not fs: print(n,"no file"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0]
        if not k.startswith("fe::"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    print("==",n)
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1].values()))[:8]:
        print(k, {a:("%.3g"%b) for a,b in v.items()})
PY
