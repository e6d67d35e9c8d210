"""-m gpu: the HIP-backed FastEGNN module (through the C ABI) against the golden vectors captured
from the reference and against the fp64 oracle (calibrated tolerances, tests/helpers.py)."""
import pytest
import torch

from tests.gpu_util import model_from_golden
from tests.helpers import Golden, golden_names, golden_loss, check_parity, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_names())
def test_forward_matches_reference_golden(name):
    g = Golden(name)
    m = model_from_golden(g)
    kw, _, _ = g.model_kwargs(device="cuda")
    with torch.no_grad():
        loc, vloc = m(**kw)
    msgs = check_parity(g, loc, vloc)
    assert not msgs, msgs


@pytest.mark.parametrize("name", golden_names())
def test_backward_matches_reference_golden(name):
    g = Golden(name)
    m = model_from_golden(g)
    kw, target, wv = g.model_kwargs(device="cuda")
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw.update(leaf)
    loc, vloc = m(**kw)
    golden_loss(loc, vloc, target, wv).backward()
    G = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    gin = {k: v.grad for k, v in leaf.items()}
    msgs = check_parity(g, loc.detach(), vloc.detach(), G, gin)
    assert not msgs, msgs
