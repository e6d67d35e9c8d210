#!/usr/bin/env python3
"""Per-unit instruction budget of the PRODUCERS of the two backward kernels from their gfx950 ISA:
edge_bwd_pc_kernel (unit = one 16-edge tile) and virt_bwd_cs_kernel (unit = one (16-node tile, channel) pair), by phase and
instruction class, with the issue-time model of tools/isa_budget.py.  The sources are compiled to device assembly (no GPU needed)
with the layer flags of the headline configuration as constants (-DFE_ISA_CONST=24) and the phase marks of the stamp builds as
comments (-DFE_ISA_MARK); the instructions between two marks are the phase's.  `--stamps FILE` joins the measured phase times of
tools/gpu_r6_stamps.sh (s_memtime ticks per unit and producer wave) to the table.

    python tools/isa_budget_bwd.py [--stamps gpurun_out/r6a/stamps.txt] > profiles/r06_bwd_instruction_budget.txt
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_budget import CLASSES, CSRC, FLAGS, ROOT, classify, parse  # noqa: E402

KERNELS = {
    "edge": ("layer_bwd.hip", "_ZN2fe18edge_bwd_pc_kernelILi3ELi2EEEvNS_11EdgeBwdArgsE"),
    "virt": ("virt_bwd.hip", "_ZN2fe18virt_bwd_cs_kernelILb0EEEvNS_10VirtCsArgsE"),
}
EDGE_PHASES = ["index + gather issue, geometry, wait for coordinates", "P + Q + rank-3 update (gathered rows arrive)", "SiLU 1 / SiLU 2 (value + derivative)",
               "operand split + recomputed product (x2)", "SiLU 3 + head dot", "degree / g_aggx rows, head adjoint, g_up", "publish (g_up, m) to ring 1",
               "g_aggm row, scaled split + WX1^T product, g_mp", "publish (g_mp, t) to ring 0", "scaled split + W2^T product, g_pre, g_d",
               "transpose tile, 16 + 1 scatter atomics, row walk (static: all 16 replicas)"]
VIRT_PHASES = ["ticket, phase / row flags", "head loads, pre-activation, SiLU 1", "operand split + V2 product (recomputed)", "SiLU 2", "operand split of v + 2 head products (recomputed)",
               "head x: SiLU, head dot, rank-1 sum, g_ux", "head X: the same", "g_np row, publish 4 tiles to ring A", "3 scaled splits + W3c^T g_np + 2 transposed head products",
               "g_vp, publish 2 tiles to ring B, scaled split + V2^T product", "g_pre: g_A read-modify-write, rank-1 sum, pools, phase counter (+ image refill when closing a phase)"]


def compile_asm(src, kernel, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"] + extra + ["-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    lines, on = [], False
    for ln in open(out):
        if ln.startswith(kernel + ":"):
            on = True
        if on:
            lines.append(ln.rstrip("\n"))
            if ln.startswith(".Lfunc_end"):   # (a kernel may hold several s_endpgm: the consumer and producer paths end separately)
                break
    os.unlink(out)
    if not lines:
        raise SystemExit(f"kernel {kernel} not found in {src}")
    return lines


def count(seq):
    c = collections.Counter()
    for it in seq:
        if it[0] == "inst":
            c[classify(it[1])] += 1
    return c


def phases_of(items):
    """instructions between consecutive marks, summed per mark number (a mark closes its phase); the producer region is what lies
    between the first and the last mark"""
    marks = [i for i, it in enumerate(items) if it[0] == "mark"]
    out = collections.OrderedDict()
    # the phase of the first mark starts at the loop head: take the instructions since the previous backward-branch target, bounded to 400
    prev = max(0, marks[0] - 400)
    for k, i in enumerate(marks):
        m = items[i][1]
        lo = prev if k == 0 else marks[k - 1]
        if k == 0:
            # from the last label before the first mark that is a loop header
            heads = [x for x in range(prev, i) if items[x][0] == "label" and "Loop Header" in items[x][2]]
            lo = heads[-1] if heads else prev
        out.setdefault(m, collections.Counter()).update(count(items[lo:i]))
    return out


def issue_cycles(c):
    other = c.get("split", 0) + c.get("valu", 0) + c.get("xlane", 0)
    return 4 * other + 8 * c.get("trans", 0) + 8 * c.get("mfma", 0)


def read_stamps(path):
    """-> {'virt': [ticks per phase], 'edge': [...]} from the output of tools/gpu_stamp_vbs.py"""
    out, cur = {"virt": [], "edge": []}, None
    if not path or not os.path.exists(path):
        return out
    for ln in open(path):
        if ln.startswith("virt_bwd_cs producers"):
            cur = "virt"
        elif ln.startswith("edge_bwd_pc producers"):
            cur = "edge"
        elif cur and re.match(r"^  \S", ln) and not ln.strip().startswith("total"):
            m = re.search(r"\s(\d+\.\d)\s+(\d+\.\d)%\s*$", ln)
            if m:
                out[cur].append(float(m.group(1)))
    return out


def table(title, names, ph, stamps, unit):
    print(title)
    hdr = f"{'phase':78s}" + "".join(f"{k:>7s}" for k in CLASSES) + f"{'issue cyc':>10s}{'stamp cyc':>10s}{'cyc/issue':>10s}"
    print(hdr)
    tot = collections.Counter()
    tot_st = 0.0
    for k, name in enumerate(names):
        c = ph.get(k, collections.Counter())
        tot.update(c)
        ic = issue_cycles(c)
        st = stamps[k] if k < len(stamps) else None
        tot_st += st or 0
        print(f"{name[:78]:78s}" + "".join(f"{c.get(x, 0):7d}" for x in CLASSES) + f"{ic:10d}" + (f"{st:10.0f}{st / max(ic, 1):10.2f}" if st else ""))
    ic = issue_cycles(tot)
    print(f"{'TOTAL per ' + unit:78s}" + "".join(f"{tot.get(x, 0):7d}" for x in CLASSES) + f"{ic:10d}" + (f"{tot_st:10.0f}{tot_st / max(ic, 1):10.2f}" if tot_st else ""))
    return tot, ic, tot_st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stamps", default=None, help="output of tools/gpu_r6_stamps.sh (measured phase times)")
    ap.add_argument("--extra", default="", help="extra compiler flags (a lever to be priced)")
    a = ap.parse_args()
    extra = ["-DFE_ISA_CONST=24", "-DFE_ISA_MARK"] + a.extra.split()
    st = read_stamps(a.stamps)
    try:
        rev = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        rev = "?"
    print("# Instruction budget of the PRODUCERS of edge_bwd_pc_kernel<GM_F16, 2> and virt_bwd_cs_kernel<false> from the gfx950 ISA, per unit and phase")
    print(f"# tools/isa_budget_bwd.py{' --extra ' + repr(a.extra) if a.extra else ''}; tree {rev} (+ working copy); flags of the headline configuration as constants (-DFE_ISA_CONST=24)")
    print("# classes as in profiles/r05_edge_fwd_instruction_budget.txt: mfma | trans = v_exp / v_rcp / v_sqrt | split = v_cvt_pk_f16 / v_fma_mix* | valu | xlane = DPP /")
    print("#   permlane / readlane | salu | branch | lds = ds_* | vmem = global_* | wait = s_waitcnt / s_nop.  issue cyc = 4 x (split + valu + xlane) + 8 x trans + 8 x mfma")
    print("#   (MI355X_MICROARCH.md constants table).  stamp cyc = s_memtime ticks (= shader cycles, ~2.1 GHz) the phase took per unit and producer wave in the")
    print("#   -DFE_STAMP build of the same tree on the cfg4 frame (tools/gpu_r6_stamps.sh; the stamps themselves slow the kernels by ~25 %).  cyc/issue = how many")
    print("#   cycles of wall time a wave spends per cycle of its own vector issue in that phase: 1 = the wave alone saturates its SIMD's issue port;")
    print("#   with two waves per SIMD anything above ~2 is time NO wave of the SIMD issues in (latency: LDS, global loads, ring flags, the matrix pipe).")
    print()
    for key, names, unit, title in (("edge", EDGE_PHASES, "16-edge tile", "edge_bwd_pc_kernel producers (csrc/layer_bwd.hip, stages.h), per 16-edge tile"),
                                    ("virt", VIRT_PHASES, "(tile, channel) unit", "virt_bwd_cs_kernel producers (csrc/virt_bwd.hip), per (16-node tile, channel) unit")):
        src, kern = KERNELS[key]
        items = parse(compile_asm(src, kern, extra))
        ph = phases_of(items)
        tot, ic, tot_st = table(title, names, ph, st[key], unit)
        print()
    print("what the recomputed forward products cost (the operands a stored pre-activation would replace):")
    print("  one f16x2 product on a resident image = operand split 40 + 24 MFMA + 16 folds + 32 ds_read_b64 = ~420 issue cycles;")
    print("  edge_bwd recomputes 2 of them per tile, virt_bwd_cs 3 per unit -- see the phase rows above for their share of the issue and of the stamped time.")


if __name__ == "__main__":
    main()
