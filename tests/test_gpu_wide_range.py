"""-m gpu: the exponent range of the product's arithmetic.  The default library runs its fp32-grade products on 2-part fp16 splits
(3 MFMAs instead of 6): operands beyond 65 504 overflow there, which the reference's plain fp32 (models/FastEGNN.py:102-119)
does not.  A drop-in must not need an environment variable for that: every forward is guarded
(fastegnn_check_finite into a host-mapped word) and a module whose pass left the range moves to the wide-range build
(libfastegnn_hip_x3.so, 3-part bf16 splits, fp32's range) and stays there.  Round 6: WITHOUT a host synchronisation -- the word is
polled at the next forward / backward; the pass that overflowed returns non-finite outputs and ZERO parameter gradients, the next
one is what fp32 gives.  FASTEGNN_RANGE_CHECK=sync re-runs the overflowing call itself (one synchronisation per forward).
FASTEGNN_WIDE_RANGE=1 starts on the wide-range build, =0 pins the f16x2 build and raises."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(scale, golden="c16_two_graphs", **env):
    e = {k: v for k, v in os.environ.items() if k != "FASTEGNN_WIDE_RANGE"}
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "tests.wide_range_runner", str(scale), golden], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_wide_range_library_matches_the_oracle_at_ordinary_and_at_huge_magnitudes():
    a = _run(1.0, FASTEGNN_WIDE_RANGE="1")
    assert a["lib"] == "libfastegnn_hip_x3.so" and a["wide"] and a["finite"] and a["err_loc"] < 1e-5 and not a["warned"], a
    b = _run(3e5, FASTEGNN_WIDE_RANGE="1")      # hidden features of ~1e5
    assert b["finite"] and b["ref_finite"] and b["err_loc"] < 1e-4, b


def test_default_policy_falls_back_by_itself_and_matches_the_oracle_beyond_fp16_range():
    a = _run(1.0)                               # ordinary magnitudes: stays on the f16x2 build, no warning
    assert a["lib"] == "libfastegnn_hip.so" and not a["wide"] and not a["warned"] and a["finite"] and a["err_loc"] < 1e-5, a
    b = _run(3e5)                               # hidden features of ~1e5: the first pass overflows (non-finite outputs, ZERO gradients),
    assert not b["first_finite"] and b["first_grads_finite"] and b["first_grads_zero"], b   # the next one runs on the wide-range build
    assert b["ref_finite"] and b["finite"] and b["wide"] and b["warned"] == 1, b
    assert b["err_loc"] < 1e-4 and b["grad_finite"] and b["err_grad_max"] < 1e-3, b


def test_sync_mode_reruns_the_overflowing_call_before_it_returns():
    b = _run(3e5, FASTEGNN_RANGE_CHECK="sync")  # round 5's behaviour, on request: the call that overflows is re-run and matches fp32
    assert b["first_finite"] and b["ref_finite"] and b["finite"] and b["wide"] and b["warned"] == 1, b
    assert b["err_loc"] < 1e-4 and b["grad_finite"] and b["err_grad_max"] < 1e-3, b


def test_forward_and_backward_do_not_synchronise_the_host():
    """the reference's one sync is data_batch[-1].item() (models/FastEGNN.py:267); this module has none --
    torch raises on any synchronising call while the debug mode is "error"."""
    import torch
    from tests.gpu_util import model_from_golden
    from tests.helpers import Golden, golden_loss
    g = Golden("c16_two_graphs")
    m = model_from_golden(g, device="cuda")
    kw, target, wv = g.model_kwargs(device="cuda")
    loc, vloc = m(**kw)                          # warm: CSR cache, host-mapped guard words, library load
    golden_loss(loc, vloc, target, wv).backward()
    torch.cuda.synchronize()
    assert m._range.mode == "deferred" and not m._range.wide
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(3):
            m.zero_grad(set_to_none=True)
            loc, vloc = m(**kw)
            golden_loss(loc, vloc, target, wv).backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loc).all()) and not m._range.peek()


def test_generic_activation_model_falls_back_to_its_own_wide_range_build():
    b = _run(3e5, golden="act_relu")
    assert not b["first_finite"] and b["first_grads_finite"], b
    assert b["ref_finite"] and b["finite"] and b["wide"] and b["warned"] == 1 and b["err_loc"] < 1e-4, b


def test_pinned_f16x2_build_raises_instead_of_returning_non_finite_outputs():
    c = _run(3e5, FASTEGNN_WIDE_RANGE="0")
    assert c["raised"] and "FASTEGNN_WIDE_RANGE" in c["raised"], c
