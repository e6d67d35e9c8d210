#!/usr/bin/env python3
"""Wait-state (hazard) checker that reads the BINARY: the gfx950 code objects of the built libraries / objects are disassembled
with llvm-objdump and every kernel's instruction stream is scanned for register hand-offs that need manually inserted wait
states.

Why: inline assembly is opaque to hipcc's hazard recognizer.  Rounds 4-5 shipped f16x2 operand splits that wrote MFMA A / B
operands with `v_fma_mixlo/hi_f16` inside asm strings and no pad; the default machine scheduler happened to leave two instructions
in between, `-amdgpu-sched-strategy=max-memory-clause` did not.  A regex over the SOURCE cannot see what the
compiler emits around an asm block; this reads what ships.

Rules (wait states = instructions issued in between; `s_nop N` counts N + 1; each applies on the straight-line stream, state is
dropped behind an unconditional branch / s_endpgm / s_setpc):
  R1  a non-MFMA VALU instruction writes a VGPR, an MFMA reads it as A, B or C            >= 2   (LLVM GCNHazardRecognizer::
                                                                                                 checkMAIHazards90A, "VALU writes vgpr -> mfma read")
  R2  a VALU instruction writes a VGPR, a DPP instruction reads it as its DPP source      >= 2   (checkDPPHazards)
  R3  a VALU instruction writes a VGPR, v_permlane16_swap / v_permlane32_swap names it    >= 2   (gfx950 permlane hazard)
  R4  a VALU instruction writes a VGPR, v_readlane / v_readfirstlane reads it             >= 1
  R5  an MFMA writes VGPRs, a non-MFMA instruction reads (or a VALU overwrites) one       >= N(mfma)
  R6  an MFMA writes VGPRs, another MFMA reads one as A or B (or as a C that is not the
      exact same register range: an accumulate chain on the same range needs none)        >= N(mfma)
      (an accumulate chain RETIRES the producer: the consuming MFMA is interlocked until the result exists, so a vector access of
      those registers behind it no longer races with the producer's write-back -- R5 then applies to the consumer's destination only)
N(mfma) is CALIBRATED: the smallest distance hipcc's own recognizer leaves for that MFMA opcode in compiler-scheduled code of the same
binary is what the hardware needs at most; the tool takes the gfx950 numbers (XDL shapes: passes + 4 for a VALU / memory reader, i.e.
8 for v_mfma_f32_16x16x32_f16 and 12 for v_mfma_f32_32x32x16_f16; the fp32-input shapes: passes + 2, i.e. 10 for
v_mfma_f32_16x16x4_f32) and `--observed` prints the minimum found beside them (equal on the round-6 tree).

    python tools/isa_hazards.py fastegnn_amd/libfastegnn_hip.so [more .so / .o / code objects]
    python tools/isa_hazards.py --build layer_fwd.hip layer_bwd.hip virt_bwd.hip misc.hip [--sched both] [--extra "-DFE_ACT_GENERIC"]
"""
from __future__ import annotations

import argparse
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile
from typing import Dict, List, Optional, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fastegnn_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
HIPCC = "/opt/rocm/bin/hipcc"
SCHED_FLAG = ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]
BASE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I../../include"]

# wait states an MFMA result needs before a non-MFMA reader / an MFMA reading it as A / B (gfx950: passes + 4 / passes + 3 + 1)
MFMA_PASSES = [
    (re.compile(r"v_mfma_f32_32x32x16_(f16|bf16)"), 8),
    (re.compile(r"v_mfma_f32_16x16x32_(f16|bf16)"), 4),
    (re.compile(r"v_mfma_f32_32x32x8_?(f16|bf16)"), 8),    # gfx942 shapes, should they ever appear
    (re.compile(r"v_mfma_f32_16x16x16_?(f16|bf16)"), 4),
    (re.compile(r"v_mfma_f32_16x16x4_?f32"), 8),
    (re.compile(r"v_mfma_f32_32x32x2_?f32"), 16),
    (re.compile(r"v_mfma_f32_4x4x"), 2),
]


def mfma_passes(op: str) -> int:
    for rx, p in MFMA_PASSES:
        if rx.match(op):
            return p
    return 16   # unknown shape: the longest pipeline


def is_xdl(op: str) -> bool:
    """the fp32-input shapes run on the legacy (non-XDL) matrix path: their results need passes + 2 states, XDL results passes + 4"""
    return not re.search(r"x\d+_?f32$", op)


def need_valu_read(op: str) -> int:
    return mfma_passes(op) + (4 if is_xdl(op) else 2)


def need_mfma_ab_read(op: str) -> int:
    return mfma_passes(op) + 4


class Inst:
    __slots__ = ("op", "dst", "src", "text", "addr", "is_mfma", "is_valu", "is_dpp", "ws", "mfma_c", "dpp_src")

    def __init__(self, op, dst, src, text, addr):
        self.op, self.dst, self.src, self.text, self.addr = op, dst, src, text, addr
        self.is_mfma = op.startswith("v_mfma") or op.startswith("v_smfma")
        self.is_valu = op.startswith("v_") and not self.is_mfma
        self.is_dpp = op.endswith("_dpp") or " quad_perm:" in text or " row_" in text and "row_mask" in text
        self.ws = 1          # wait states this instruction contributes
        self.mfma_c = None   # (lo, hi) of the C operand of an MFMA, when it is a register range
        self.dpp_src = []    # VGPRs a DPP instruction reads THROUGH the lane crossbar (its first source operand)


_VREG = re.compile(r"(?<![a-z0-9_])v(\d+)(?![0-9:\[])|v\[(\d+):(\d+)\]")


def vregs(operand: str) -> List[int]:
    out = []
    for m in _VREG.finditer(operand):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(rest: str) -> List[str]:
    """operands of an instruction line (modifiers like `op_sel:[1,0,0]` stay attached to the text, they name no register)"""
    rest = re.sub(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi|quad_perm|cbsz|abid|blgp)\s*:\s*\[[^\]]*\]", "", rest)
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


# instructions whose FIRST operand is not a VGPR destination written by the VALU
_NO_VDST = re.compile(r"v_(cmp|cmpx|readlane|readfirstlane|nop)")
_TWO_DST = re.compile(r"v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|div_scale|mad_u64_u32|mad_i64_i32)")


def parse_line(line: str) -> Optional[Inst]:
    s = line.split("//")[0].strip()
    if not s or s.startswith(";") or s.endswith(":") or s.startswith("."):
        return None
    parts = s.split(None, 1)
    op = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    if not re.match(r"^[a-z_0-9]+$", op):
        return None
    ops = split_operands(rest)
    dst: List[int] = []
    src: List[int] = []
    m = re.search(r"//\s*([0-9A-Fa-f]+):", line)
    addr = m.group(1) if m else ""
    ins = Inst(op, dst, src, s, addr)
    if op == "s_nop":
        try:
            ins.ws = int(ops[0], 0) + 1
        except Exception:
            ins.ws = 1
        return ins
    if op.startswith("v_"):
        if _NO_VDST.match(op):
            for o in ops[1:] if not op.startswith("v_cmpx") and not op.startswith("v_cmp") else ops:
                src.extend(vregs(o))
            if op.startswith("v_cmp"):   # v_cmp* / v_cmpx*: every VGPR named is a source
                src[:] = []
                for o in ops:
                    src.extend(vregs(o))
            return ins
        if "permlane" in op and "swap" in op:   # both operands are read and written
            for o in ops[:2]:
                dst.extend(vregs(o))
                src.extend(vregs(o))
            return ins
        if ops:
            dst.extend(vregs(ops[0]))
        rest_ops = ops[1:]
        if _TWO_DST.match(op) and rest_ops:
            rest_ops = rest_ops[1:]           # second destination is an SGPR pair / vcc
        for o in rest_ops:
            src.extend(vregs(o))
        if ins.is_dpp and rest_ops:
            ins.dpp_src = vregs(rest_ops[0])
        if ins.is_mfma and len(ops) >= 4:
            c = vregs(ops[3])
            if c:
                ins.mfma_c = (min(c), max(c))
        # read-modify-write destinations (half-register writes, accumulating forms): the old value is a source too
        if re.match(r"v_(fma_mixlo|fma_mixhi|mad_mixlo|mad_mixhi|fmac|mac|dot\w*c|pk_fmac|cvt_scalef32_pk|cvt_sr)", op) or " dst_sel:" in s or ins.is_dpp and "bound_ctrl" not in s:
            src.extend(dst)
        return ins
    # memory / LDS / scalar instructions: loads write VGPRs through the memory pipeline (covered by s_waitcnt, not by wait states);
    # every other VGPR named is read at issue
    is_load = bool(re.match(r"(global_load|buffer_load|flat_load|scratch_load|ds_read|ds_load|ds_bpermute|ds_permute|ds_swizzle|ds_consume|ds_append|"
                            r"global_atomic\w*|buffer_atomic\w*|flat_atomic\w*|ds_\w*_rtn|image_)", op)) and "lds" not in op.split("_")[-1:]
    if is_load and ops:
        returns = not (re.match(r"(global|buffer|flat)_atomic", op) and " sc0" not in s and " glc" not in s)
        if returns:
            for o in ops[1:]:
                src.extend(vregs(o))
            return ins
    for o in ops:
        src.extend(vregs(o))
    return ins


def parse_kernels(dis_text: str) -> Dict[str, List[Inst]]:
    kernels: Dict[str, List[Inst]] = collections.OrderedDict()
    cur = None
    for line in dis_text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:\s*$", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None:
            continue
        ins = parse_line(line)
        if ins is not None:
            kernels[cur].append(ins)
    return kernels


FAR = 1 << 20


def check_kernel(insts: List[Inst], observed: Optional[dict] = None) -> List[Tuple[str, str, Inst, Inst, int, int]]:
    """-> [(rule, register, producer, consumer, wait states found, needed)]"""
    # last writer per VGPR: (position in wait states, Inst)
    w_valu: Dict[int, Tuple[int, Inst]] = {}
    w_mfma: Dict[int, Tuple[int, Inst]] = {}
    pos = 0
    out = []
    retired = set()

    def since(entry):
        return pos - entry[0] - 1   # wait states strictly between the two instructions (the producer itself took one slot)

    for ins in insts:
        if ins.op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
            w_valu.clear()
            w_mfma.clear()
            pos += ins.ws
            continue
        # ---- consumer side ----
        if ins.is_mfma:
            for r in set(ins.src):
                e = w_valu.get(r)
                if e is not None and since(e) < 2:
                    out.append(("R1", f"v{r}", e[1], ins, since(e), 2))
                e = w_mfma.get(r)
                if e is not None:
                    prod = e[1]
                    same_chain = ins.mfma_c is not None and prod.dst and ins.mfma_c == (min(prod.dst), max(prod.dst)) and \
                        ins.mfma_c[0] <= r <= ins.mfma_c[1] and r not in _ab_regs(ins)
                    if same_chain:
                        retired.add(r)
                    if not same_chain:
                        need = need_mfma_ab_read(prod.op)
                        if observed is not None:
                            k = ("mfma->mfma", prod.op)
                            observed[k] = min(observed.get(k, FAR), since(e))
                        if since(e) < need:
                            out.append(("R6", f"v{r}", prod, ins, since(e), need))
        else:
            reads = set(ins.src)
            if ins.is_dpp:
                for r in set(ins.dpp_src):
                    e = w_valu.get(r)
                    if e is not None and since(e) < 2:
                        out.append(("R2", f"v{r}", e[1], ins, since(e), 2))
            if "permlane" in ins.op and "swap" in ins.op:
                for r in reads:
                    e = w_valu.get(r)
                    if e is not None and since(e) < 2:
                        out.append(("R3", f"v{r}", e[1], ins, since(e), 2))
            if ins.op.startswith("v_readlane") or ins.op.startswith("v_readfirstlane"):
                for r in reads:
                    e = w_valu.get(r)
                    if e is not None and since(e) < 1:
                        out.append(("R4", f"v{r}", e[1], ins, since(e), 1))
            touched = reads | (set(ins.dst) if ins.is_valu else set())
            for r in touched:
                e = w_mfma.get(r)
                if e is not None:
                    need = need_valu_read(e[1].op)
                    if observed is not None:
                        k = ("mfma->other", e[1].op)
                        observed[k] = min(observed.get(k, FAR), since(e))
                    if since(e) < need:
                        out.append(("R5", f"v{r}", e[1], ins, since(e), need))
        # An MFMA that takes a whole earlier result as its C operand (an accumulate chain) is interlocked by the hardware until that
        # result exists: behind it the earlier MFMA's write-back is done, and a later vector access of those registers races with
        # nothing (LLVM's static count would still pad it; the elapsed time is >= the producer's passes).  The consumer's own
        # destination is tracked as usual.
        for r in retired:
            w_mfma.pop(r, None)
        retired.clear()
        # ---- producer side ----
        if ins.is_mfma:
            for r in ins.dst:
                w_mfma[r] = (pos, ins)
                w_valu.pop(r, None)
        elif ins.is_valu:
            for r in ins.dst:
                w_valu[r] = (pos, ins)
                w_mfma.pop(r, None)
        else:
            # a memory load's destination is rewritten by the memory pipeline: older VALU / MFMA writers no longer matter
            if ins.op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "ds_read", "ds_load")):
                ops = split_operands(ins.text.split(None, 1)[1] if " " in ins.text else "")
                if ops:
                    for r in vregs(ops[0]):
                        w_valu.pop(r, None)
                        w_mfma.pop(r, None)
        pos += ins.ws
        # forget writers that are out of every window
        if len(w_valu) > 512:
            for r in [r for r, e in w_valu.items() if pos - e[0] > 64]:
                del w_valu[r]
        if len(w_mfma) > 512:
            for r in [r for r, e in w_mfma.items() if pos - e[0] > 64]:
                del w_mfma[r]
    return out


def _ab_regs(ins: Inst) -> set:
    ops = split_operands(ins.text.split(None, 1)[1]) if " " in ins.text else []
    s = set()
    for o in ops[1:3]:
        s.update(vregs(o))
    return s


def code_objects_of(path: str, workdir: str) -> List[str]:
    """gfx950 code objects inside `path` (a host .so / .o with offload bundles, or a code object itself)"""
    with open(path, "rb") as f:
        head = f.read(20)
    base = os.path.join(workdir, os.path.basename(path))
    shutil.copy(path, base)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":   # e_machine EM_AMDGPU (224)
        return [base]
    if head.startswith(b"__CLANG_OFFLOAD_BUND"):                   # a bare offload bundle (`hipcc --cuda-device-only -c`)
        out = base + ".gfx950.co"
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + base,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return [out]
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.basename(base)], cwd=workdir, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(glob.glob(base + ".*gfx950*"))


def disassemble(code_object: str) -> str:
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", code_object], check=True, capture_output=True, text=True).stdout


def check_file(path: str, only: Optional[re.Pattern] = None, observed: Optional[dict] = None):
    """-> (n kernels, n instructions, violations [(kernel, rule, reg, producer text, consumer text, found, needed)])"""
    viol, nk, ni = [], 0, 0
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects_of(path, wd):
            for name, insts in parse_kernels(disassemble(co)).items():
                if only is not None and not only.search(name):
                    continue
                nk += 1
                ni += len(insts)
                for rule, reg, p, c, found, need in check_kernel(insts, observed):
                    viol.append((name, rule, reg, f"{p.addr}: {p.text}", f"{c.addr}: {c.text}", found, need))
    return nk, ni, viol


def build_objects(files: List[str], sched: bool, extra: List[str], outdir: str) -> List[str]:
    outs = []
    for f in files:
        o = os.path.join(outdir, os.path.splitext(f)[0] + (".mmc" if sched else ".def") + ".co")
        cmd = [HIPCC] + BASE_FLAGS + (SCHED_FLAG if sched else []) + extra + ["--cuda-device-only", "-c", f, "-o", o]
        subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        outs.append(o)
    return outs


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("paths", nargs="*", help=".so / .o / code objects; with --build: .hip files of fastegnn_amd/csrc")
    ap.add_argument("--build", action="store_true", help="compile the named csrc/*.hip files (device only) and check them")
    ap.add_argument("--sched", choices=["default", "mmc", "both"], default="both", help="--build: machine-scheduler strategies to compile under")
    ap.add_argument("--extra", default="", help="--build: extra compiler flags, e.g. '-DFE_ACT_GENERIC'")
    ap.add_argument("--only", default=None, help="regex on the (mangled) kernel name")
    ap.add_argument("--observed", action="store_true", help="print the smallest MFMA-result distances seen (calibration of R5 / R6)")
    a = ap.parse_args()
    only = re.compile(a.only) if a.only else None
    targets = []
    tmp = None
    if a.build:
        tmp = tempfile.mkdtemp(prefix="isa_haz_")
        for mode in (["default", "mmc"] if a.sched == "both" else [a.sched]):
            targets += build_objects(a.paths, mode == "mmc", a.extra.split(), tmp)
    else:
        targets = a.paths or sorted(glob.glob(os.path.join(ROOT, "fastegnn_amd", "libfastegnn_hip*.so")))
    bad = 0
    observed = {} if a.observed else None
    for t in targets:
        nk, ni, viol = check_file(t, only, observed)
        print(f"{os.path.basename(t)}: {nk} kernels, {ni} instructions, {len(viol)} hazard violations")
        for v in viol[:40]:
            print(f"  {v[1]} {v[2]} in {v[0][:70]}: {v[5]} wait states, needs {v[6]}\n      producer {v[3]}\n      consumer {v[4]}")
        bad += len(viol)
    if observed:
        for (kind, op), d in sorted(observed.items()):
            print(f"observed minimum {kind:12s} {op:34s} {d} wait states")
    if tmp:
        shutil.rmtree(tmp, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
