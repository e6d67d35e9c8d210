"""profiles/r05_wide_path.txt from the files tools/gpu_r5_wide.sh leaves under gpurun_out/wide_r5/ (report.txt, kernel_stats.csv):
the step times, the per-operator table and the rocprofv3 kernel table per step with the kernel families' shares."""
import csv
import re
import sys

extra = sys.argv[1] if len(sys.argv) > 1 else ""
rep = open('gpurun_out/wide_r5/report.txt').read()
if 'rocprofv3 --kernel-trace --stats of:' in rep:
    rep = rep[:rep.index('rocprofv3 --kernel-trace --stats of:')]
rows = list(csv.DictReader(open('gpurun_out/wide_r5/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = [rep.rstrip(), ""]
if extra:
    out += [extra, ""]
out += ["rocprofv3 --kernel-trace --stats of: python3 tools/gpu_wide_timing.py 100000 16 128   (4 steps; per step)",
        f"{'kernel':84s} {'calls':>6s} {'ms/step':>8s} {'avg us':>9s} {'%':>6s}"]
fam = {}
for r in rows[:26]:
    name = re.sub(r'\(.*', '', r['Name']).replace('void ', '')[:84]
    out.append(f"{name:84s} {int(r['Calls']) // 4:6d} {float(r['TotalDurationNs']) / 4e6:8.2f} {float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")
for r in rows:
    n = r['Name']
    k = ('GEMM family (gemm_x3 + tn_x3)' if ('gemm_x3' in n or 'tn_x3' in n) else 'run scatters + gathers' if ('scatter_add' in n or 'gather2' in n)
         else 'narrow GEMMs / column sums' if ('smalln' in n or 'smallk' in n or 'tn_small' in n or 'colsum' in n)
         else 'torch elementwise / fills / sorts' if ('at::native' in n or 'rocprim' in n or 'rocclr' in n) else 'other fe:: kernels')
    fam[k] = fam.get(k, 0) + float(r['TotalDurationNs'])
out += ["", f"by family (of {tot / 4e6:.1f} ms of kernel time per step):"]
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    out.append(f"  {k:40s} {v / 4e6:7.2f} ms  {100 * v / tot:5.1f} %")
open('profiles/r05_wide_path.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out[-12:]))
