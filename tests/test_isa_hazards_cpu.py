"""CPU: the wait-state (hazard) checker on the BINARY (tools/isa_hazards.py).

Inline assembly is opaque to hipcc's hazard recognizer.  Rounds 4-5 shipped f16x2 operand splits whose asm blocks wrote MFMA A / B
operands without the two VALU -> MFMA wait states; 419 GPU tests passed because the default machine scheduler happened to leave two
instructions in between.  These tests read what the compiler EMITTED:
  * every product library in the tree (what ships to the GPU box and what the round-end run maps) has no hand-off below its wait
    states, over every kernel (also the wide path's);
  * the checker itself is pinned on hand-written instruction streams, one per rule;
  * a build of csrc/layer_fwd.hip without the pad of common.h: mix_pack4 (-DFE_HAZARD_SELFTEST, with the round-5 order of the f16x2
    products: -DFE_F2_PIPE_FWD=0) under the max-memory-clause scheduler strategy -- the exact tree round 5 found broken -- is
    reported; the same build with the pad is clean.
Round 6's own catch with this checker: an "=&v" output of the operand-split asm block allocated to a register that an MFMA had written
two instructions earlier (R5; common.h: mix_pack4<IN_PLACE>).
"""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards as HZ  # noqa: E402

LIBS = sorted(glob.glob(os.path.join(ROOT, "fastegnn_amd", "libfastegnn_hip*.so")))
HAVE_TOOLS = os.path.exists(os.path.join(HZ.LLVM, "llvm-objdump")) and os.path.exists(HZ.HIPCC)
pytestmark = pytest.mark.skipif(not HAVE_TOOLS, reason="needs hipcc + llvm-objdump (the build container has them)")


def _kernel(text):
    return [i for i in (HZ.parse_line("\t" + ln.strip()) for ln in text.strip().splitlines()) if i is not None]


def _rules(text):
    return sorted({v[0] for v in HZ.check_kernel(_kernel(text))})


def test_checker_rules_on_handwritten_streams():
    # R1: VALU write -> MFMA operand, fewer than two wait states in between
    assert _rules("""
        v_fma_mixhi_f16 v62, v67, v161, 0
        v_mfma_f32_16x16x32_f16 v[18:21], v[66:69], v[62:65], v[18:21]""") == ["R1"]
    assert _rules("""
        v_fma_mixhi_f16 v62, v67, v161, 0
        s_nop 0
        v_mfma_f32_16x16x32_f16 v[18:21], v[66:69], v[62:65], v[18:21]""") == ["R1"]
    assert _rules("""
        v_fma_mixhi_f16 v62, v67, v161, 0
        s_nop 1
        v_mfma_f32_16x16x32_f16 v[18:21], v[66:69], v[62:65], v[18:21]""") == []
    assert _rules("""
        v_cvt_pk_f16_f32 v62, v1, v2
        ds_read_b64 v[66:67], v149
        ds_read_b64 v[68:69], v149 offset:32
        v_mfma_f32_16x16x32_f16 v[18:21], v[66:69], v[62:65], v[18:21]""") == []
    # the C operand counts too
    assert _rules("""
        v_mov_b32_e32 v18, 0
        v_mfma_f32_16x16x32_f16 v[18:21], v[66:69], v[62:65], v[18:21]""") == ["R1"]
    # R2 / R3: VALU write -> DPP source / permlane swap operand
    assert _rules("""
        v_add_f32_e32 v5, v1, v2
        v_add_f32_dpp v7, v5, v5 row_shl:8 row_mask:0xf bank_mask:0x3""") == ["R2"]
    assert _rules("""
        v_add_f32_e32 v5, v1, v2
        s_nop 1
        v_add_f32_dpp v7, v5, v5 row_shl:8 row_mask:0xf bank_mask:0x3""") == []
    assert _rules("""
        v_mov_b32_e32 v5, v1
        v_permlane16_swap_b32_e32 v5, v6""") == ["R3"]
    assert _rules("""
        v_mov_b32_e32 v5, v1
        s_nop 1
        v_permlane16_swap_b32_e32 v5, v6""") == []
    # R5: an MFMA result read by a vector instruction before it has left the pipe (16x16x32: 8 states, 32x32x16: 12)
    assert _rules("""
        v_mfma_f32_16x16x32_f16 v[74:77], v[70:73], v[62:65], v[74:77]
        s_nop 6
        v_fmamk_f32 v18, v74, 0x3a000000, v18""") == ["R5"]
    assert _rules("""
        v_mfma_f32_16x16x32_f16 v[74:77], v[70:73], v[62:65], v[74:77]
        s_nop 7
        v_fmamk_f32 v18, v74, 0x3a000000, v18""") == []
    assert _rules("""
        v_mfma_f32_32x32x16_f16 v[0:15], v[70:73], v[62:65], v[0:15]
        s_nop 7
        s_nop 2
        v_mul_f32_e32 v20, v3, v3""") == ["R5"]
    # an accumulate chain on the same registers needs nothing; the result as the A operand of the next MFMA does (R6)
    assert _rules("""
        v_mfma_f32_16x16x32_f16 v[74:77], v[70:73], v[62:65], v[74:77]
        v_mfma_f32_16x16x32_f16 v[74:77], v[66:69], v[58:61], v[74:77]""") == []
    assert _rules("""
        v_mfma_f32_16x16x32_f16 v[74:77], v[70:73], v[62:65], v[74:77]
        v_mfma_f32_16x16x32_f16 v[18:21], v[74:77], v[58:61], v[18:21]""") == ["R6"]
    # ... and retires the producer: its registers may be rewritten behind the consuming MFMA (the low-part accumulator of an f16x2 product
    # is dead once the next MFMA has taken it as C, and the allocator hands it to the next asm block)
    assert _rules("""
        v_mfma_f32_16x16x32_f16 v[12:15], v[88:91], v[8:11], v[12:15]
        v_mfma_f32_16x16x32_f16 v[84:87], v[84:87], v[0:3], v[12:15]
        v_fma_mixlo_f16 v12, v16, v55, 0""") == []
    assert _rules("""
        v_mfma_f32_16x16x32_f16 v[12:15], v[88:91], v[8:11], v[12:15]
        v_fma_mixlo_f16 v12, v16, v55, 0""") == ["R5"]
    # state is dropped behind an unconditional branch
    assert _rules("""
        v_fma_mixhi_f16 v62, v67, v161, 0
        s_branch 12
        v_mfma_f32_16x16x32_f16 v[18:21], v[66:69], v[62:65], v[18:21]""") == []


@pytest.mark.skipif(not LIBS, reason="no built library in the tree (run __graft_entry__.build())")
@pytest.mark.parametrize("lib", LIBS, ids=[os.path.basename(p) for p in LIBS])
def test_shipped_library_has_no_hazard_violation(lib):
    nk, ni, viol = HZ.check_file(lib)
    assert nk >= 40 and ni > 100000, (nk, ni)          # every kernel of the library was read (the stage kernels + the wide path)
    assert not viol, viol[:5]


def _device_object(tmp_path, name, extra):
    out = str(tmp_path / (name + ".co"))
    cmd = [HZ.HIPCC] + HZ.BASE_FLAGS + HZ.SCHED_FLAG + extra + ["--cuda-device-only", "-c", "layer_fwd.hip", "-o", out]
    subprocess.run(cmd, cwd=HZ.CSRC, check=True, stderr=subprocess.DEVNULL)
    return out


def test_a_tree_without_the_operand_split_pad_fails_the_check(tmp_path):
    """the round-5 bug, reproduced: mix_pack4 without `s_nop 1` under -amdgpu-sched-strategy=max-memory-clause"""
    bad = _device_object(tmp_path, "nopad", ["-DFE_HAZARD_SELFTEST", "-DFE_F2_PIPE_FWD=0"])
    nk, ni, viol = HZ.check_file(bad)
    r1 = [v for v in viol if v[1] == "R1"]
    assert r1, "the checker did not see the missing VALU -> MFMA wait states"
    assert any("v_fma_mix" in v[3] for v in r1), r1[:3]   # the producer is the asm-written packed half
    good = _device_object(tmp_path, "pad", ["-DFE_F2_PIPE_FWD=0"])
    assert HZ.check_file(good)[2] == []
