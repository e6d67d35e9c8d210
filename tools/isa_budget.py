#!/usr/bin/env python3
"""Per-tile instruction budget of edge_fwd_kernel from its gfx950 ISA.

Compiles csrc/layer_fwd.hip to device assembly twice (no GPU needed: hipcc cross-compiles) with the layer flags of the
headline configuration as compile-time constants (-DFE_ISA_CONST: no attention / normalize / tanh, edge_attr_nf = 2), so the
16-edge tile is straight-line code:
  * the SHIPPED schedule (no marks): instruction totals of one trip of the tile loop -- what SQ_INSTS_* count;
  * the same with -DFE_ISA_MARK: the FE_T(i) phase boundaries of stages.h / layer_fwd.hip become comments between scheduling
    fences, and the instructions between two marks are the phase's.
Every instruction is classed by opcode (transcendental, f16x2 operand split, MFMA, plain VALU, cross-lane, SALU, LDS, VMEM,
waits).  The row walk that ends a tile is data dependent (one branch pair per edge, a flush per row change): its cost is
modelled as 16 x the no-change path + (16 / mean degree) x the change path.

    python tools/isa_budget.py [--degree 19.2] [--extra "-DFE_..."] > profiles/r05_edge_fwd_instruction_budget.txt
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fastegnn_amd", "csrc")
KERNEL = "_ZN2fe15edge_fwd_kernelILi3EEEvNS_8EdgeArgsEi"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I../../include"]

CLASSES = ["mfma", "trans", "split", "valu", "xlane", "salu", "branch", "lds", "vmem", "wait"]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if re.match(r"v_(exp|rcp|rsq|sqrt|log|sin|cos)_", op):
        return "trans"
    if re.match(r"v_(cvt_pk_f16|cvt_pk_bf16|cvt_pkrtz|fma_mix|perm_b32)", op):
        return "split"
    if re.match(r"v_(readlane|readfirstlane|writelane|permlane|mov_b32_dpp|add_f32_dpp)", op) or op.endswith("_dpp"):
        return "xlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op in ("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if re.match(r"(global|buffer|flat|scratch)_", op):
        return "vmem"
    return "valu"


def compile_asm(extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", "layer_fwd.hip", "-o", out]
    subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    lines, on = [], False
    for ln in open(out):
        if ln.startswith(KERNEL + ":"):
            on = True
        if on:
            lines.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    os.unlink(out)
    return lines


def parse(lines):
    """-> list of (kind, text): kind in {'label', 'mark', 'inst'}"""
    items = []
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(";;") or s.startswith(".") and not s.startswith(".LBB"):
            continue
        m = re.match(r"; FE_MARK (\d+)", s)
        if m:
            items.append(("mark", int(m.group(1)), s))
            continue
        if s.startswith(";"):
            continue
        if s.startswith(".LBB"):
            items.append(("label", s.split(":")[0], s))
            continue
        if s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"[a-z]", op):
            continue
        items.append(("inst", op, s))
    return items


def tile_loop(items):
    """The tile loop = the outermost depth-1 loop that contains a row-walk (v_readlane) -- located by its header label and the
    LAST backward branch to it."""
    heads = [i for i, it in enumerate(items) if it[0] == "label" and "Loop Header: Depth=1" in it[2] and "Inner" not in it[2]]
    best = None
    for h in heads:
        name = items[h][1]
        back = [i for i, it in enumerate(items) if it[0] == "inst" and (it[1].startswith("s_cbranch") or it[1] == "s_branch") and it[2].split()[-1] == name and i > h]
        if back and any(it[0] == "inst" and it[1].startswith("v_mfma") for it in items[h:back[-1]]):
            best = (h, back[-1])
    if best is None:
        raise SystemExit("tile loop not found")
    return best


def count(seq):
    c = collections.Counter()
    for it in seq:
        if it[0] == "inst":
            c[classify(it[1])] += 1
    return c


def fmt_row(name, c, width=44):
    return f"{name:{width}s}" + "".join(f"{c.get(k, 0):8.1f}" for k in CLASSES) + f"{sum(c.get(k, 0) for k in ('mfma','trans','split','valu','xlane')):9.1f}"


def walk_model(items, lo, hi, degree):
    """Row walk between FE_MARK 5 and FE_MARK 6.  The compiler replicates the code of one edge sixteen times:
         s_cmp_lt (edge < nvalid) + branch, s_bitcmp0 (row starts here?) + branch, [change path], adds of the two columns.
    head      = everything before the first of these tests (the 32 LDS column reads, the DPP compare + ballot),
    change    = the instructions between the s_bitcmp0 branch of edge 0 and the label it jumps to (flush of the finished row: 1/deg,
                scale, two stores; the zero_rows loop, which runs zero times unless a row has no edges),
    no-change = the two tests + the instructions from that label to the test of edge 1 (the two adds, the edge counter).
    Per tile: head + 16 x no-change + (16 / mean degree) x change."""
    seq = items[lo:hi]
    tests = [i for i, it in enumerate(seq) if it[0] == "inst" and it[1] == "s_bitcmp0_b32"]
    # head: up to the first scalar test of the walk
    first_cmp = next(i for i, it in enumerate(seq) if it[0] == "inst" and it[1] == "s_cmp_lt_i32")
    head = count(seq[:first_cmp])
    head.update(count([it for it in seq[first_cmp:] if it[0] == "inst" and it[1].startswith("ds_read")]))   # column reads placed inside replicas
    # the replica of edge 1 (edge 0's is fused with the walk's head): s_bitcmp0 + branch -> [change path] -> join: adds, counter,
    # the nvalid test of the next edge + branch
    t1 = tests[1]
    br = next(i for i in range(t1, len(seq)) if seq[i][0] == "inst" and seq[i][1].startswith("s_cbranch"))
    target = seq[br][2].split()[-1]
    join = next(i for i in range(br, len(seq)) if seq[i][0] == "label" and seq[i][1] == target)
    change = count(seq[br + 1:join])
    end = next(i for i in range(join, len(seq)) if seq[i][0] == "inst" and seq[i][1].startswith("s_cbranch"))
    nochange = count(seq[t1:br + 1]) + count(seq[join:end + 1])
    per_tile = collections.Counter()
    for k in CLASSES:
        per_tile[k] = head.get(k, 0) + 16 * nochange.get(k, 0) + (16.0 / degree) * change.get(k, 0)
    return head, nochange, change, per_tile


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=float, default=19.19, help="mean in-degree of the frame (cfg4: 1 919 172 / 100 000)")
    ap.add_argument("--extra", default="", help="extra compiler flags (a lever to be priced)")
    args = ap.parse_args()
    extra = args.extra.split()
    plain = parse(compile_asm(["-DFE_ISA_CONST=24"] + extra))
    marked = parse(compile_asm(["-DFE_ISA_CONST=24", "-DFE_ISA_MARK"] + extra))

    h, b = tile_loop(plain)
    total_static = count(plain[h:b + 1])
    hm, bm = tile_loop(marked)
    marks = [(i, it[1]) for i, it in enumerate(marked) if it[0] == "mark" and hm <= i <= bm]
    names = {0: "index + gather issue, geometry (d, r, sqrt)", 1: "first layer: P + Q + rank-3 update (4 fp32 MFMA)",
             2: "SiLU", 3: "operand split (f16x2) + 64x64 product", 4: "SiLU 3 + coordinate head dot (qsum)",
             5: "transpose tile to LDS (m, x*s)"}
    phases = collections.OrderedDict()
    prev = hm
    order = []
    for i, m in marks:
        key = (m, sum(1 for k in order if k[0] == m))
        order.append(key)
        phases[key] = count(marked[prev:i])
        prev = i
    # after the last mark 5: the row walk up to the back edge
    last5 = max(i for i, m in marks if m == 5)
    # marks after last5 inside the loop (FE_MARK 6 sits at the end of the walk)
    end_walk = next((i for i, m in marks if i > last5), bm)
    end6 = next((i for i, m in marks if i > last5 and m == 6), bm + 1)
    head, nochange, change, walk = walk_model(marked, last5, end6, args.degree)

    print("# edge_fwd_kernel<GM_F16> (csrc/layer_fwd.hip, stages.h): instructions per 16-edge tile, from the gfx950 ISA")
    print("# tools/isa_budget.py" + (f" --extra '{args.extra}'" if args.extra else "") + f"; mean degree {args.degree}")
    try:
        rev = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        rev = "?"
    print(f"# tree: {rev} (+ working copy); flags of the headline configuration as constants (-DFE_ISA_CONST=24, edge_attr_nf = 2)")
    print("# classes: mfma | trans = v_exp / v_rcp / v_sqrt (8 issue cycles each, plain vector instructions 4) | split = v_cvt_pk_f16 /")
    print("#          v_fma_mix* (the f16x2 operand split) | valu = every other vector instruction | xlane = DPP / permlane / readlane |")
    print("#          salu | branch | lds = ds_* | vmem = global_* | wait = s_waitcnt / s_nop;  'vector' = mfma + trans + split + valu + xlane")
    print()
    hdr = f"{'phase (marks of stages.h / layer_fwd.hip)':44s}" + "".join(f"{k:>8s}" for k in CLASSES) + f"{'vector':>9s}"
    print(hdr)
    tot = collections.Counter()
    seen = collections.Counter()
    for (m, k), c in phases.items():
        if m == 6:      # the static row walk (all sixteen replicas): replaced by the model below
            continue
        seen[m] += 1
        label = names.get(m, f"mark {m}")
        if m in (2, 3):
            label += f" #{k + 1}"
        print(fmt_row(label, c))
        tot.update(c)
    print(fmt_row("row walk + segmented sums (modelled)", walk))
    tot.update(walk)
    print(fmt_row("TOTAL per tile (marked build)", tot))
    print()
    print("row walk model: head (the LDS column reads, DPP compare + ballot) " + dict(head).__repr__())
    print("                per edge, no row change " + dict(nochange).__repr__())
    print("                per row change (flush: 1/deg, two stores, zero_rows test) " + dict(change).__repr__())
    print()
    print("shipped schedule (no marks), ONE static trip of the tile loop with every row-walk replica counted once")
    print("(an over-count of the walk: all 16 change paths are in it):")
    print(fmt_row("static loop body", total_static))
    # by-purpose roll-up
    silu = sum((phases[k] for k in phases if k[0] == 2), collections.Counter())
    gemm = sum((phases[k] for k in phases if k[0] == 3), collections.Counter())
    print()
    print("roll-up by purpose (vector-pipe instructions per tile; share of the marked total):")
    vec = lambda c: sum(c.get(k, 0) for k in ("mfma", "trans", "split", "valu", "xlane"))
    T = vec(tot)
    rows = [("geometry + index / gather issue", vec(phases.get((0, 0), {}))),
            ("first-layer sum (P + Q + rank-3 MFMA update)", vec(phases.get((1, 0), {}))),
            ("SiLU 1 + SiLU 2 (48 + 48 transcendental)", vec(silu)),
            ("operand splits (2 x 16 elements x 2.5) + 2 x 24 MFMA + folds", vec(gemm)),
            ("SiLU 3 + head dot + x update", vec(phases.get((4, 0), {}))),
            ("transpose stores", vec(phases.get((5, 0), {}))),
            ("row walk + segmented sums", vec(walk))]
    for n, v in rows:
        print(f"  {n:62s} {v:7.1f}  {100 * v / T:5.1f} %")
    print(f"  {'total':62s} {T:7.1f}")
    print()
    print("issue-time model (MI355X_MICROARCH.md, constants table 'vector-instruction ISSUE cost': v_add / v_fma / v_cvt_pk 4 cycles,")
    print("v_exp / v_rcp / v_sqrt 8, an MFMA holds the SIMD's vector issue for 8 of its 16 cycles; costs add):")
    n_trans = tot.get("trans", 0)
    n_mfma = tot.get("mfma", 0)
    n_other = tot.get("split", 0) + tot.get("valu", 0) + tot.get("xlane", 0)
    cyc = 4 * n_other + 8 * n_trans + 8 * n_mfma
    print(f"  plain vector {n_other:.0f} x 4 + transcendental {n_trans:.0f} x 8 + MFMA {n_mfma:.0f} x 8 = {cyc:.0f} issue cycles per tile and wave")
    print(f"  cfg4: 119 949 tiles per launch / 1024 SIMDs = 117.1 tiles per SIMD -> {cyc * 117.1 / 1e3:.0f} k cycles = {cyc * 117.1 / 1.66e9 * 1e6:.0f} us at 1.66 GHz")
    print(f"  the 0.40-of-HBM-peak target is 184.8 us per launch (591.4 MB / 3.2 TB/s) = {184.8e-6 * 1.66e9 / 117.1:.0f} issue cycles per tile")


if __name__ == "__main__":
    main()
