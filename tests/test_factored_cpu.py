"""CPU: the stage-level spec (oracle/factored.py: factorised forward + hand-derived backward,
the math the HIP kernels implement) against autograd of the op-for-op oracle, in fp64, and
against the reference goldens in fp32."""
import pytest
import torch

from oracle import fastegnn_ref as R
from oracle import factored as F
from tests.helpers import Golden, golden_names, golden_loss, rel_err, check_parity


def _run_pair(g, dt):
    p = {k: v.clone().requires_grad_(True) for k, v in g.tensors(g.params, dtype=dt).items()}
    kw, target, wv = g.model_kwargs(dtype=dt)
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw_ref = dict(kw); kw_ref.update(leaf)
    loc, vloc = R.forward(p, g.cfg, **kw_ref)
    golden_loss(loc, vloc, target, wv).backward()

    pd = {k: v.detach() for k, v in p.items()}
    loc2, vloc2, ctx = F.model_forward(pd, g.cfg, **kw)
    l2 = loc2.clone().requires_grad_(True); v2 = vloc2.clone().requires_grad_(True)
    golden_loss(l2, v2, target, wv).backward()
    G, gin = F.model_backward(pd, g.cfg, ctx, l2.grad, v2.grad)
    return p, leaf, loc, vloc, loc2, vloc2, G, gin


@pytest.mark.parametrize("name", golden_names(silu_only=True))
def test_factored_fp64_matches_autograd(name):
    g = Golden(name)
    p, leaf, loc, vloc, loc2, vloc2, G, gin = _run_pair(g, torch.float64)
    assert rel_err(loc2, loc) < 1e-12
    assert rel_err(vloc2, vloc) < 1e-12
    for k, v in p.items():
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        assert rel_err(G[k], gr) < 1e-9, k
    for k, v in leaf.items():
        assert rel_err(gin[k], v.grad) < 1e-9, k


@pytest.mark.parametrize("name", golden_names(silu_only=True))
def test_factored_fp32_matches_reference_golden(name):
    g = Golden(name)
    p, leaf, loc, vloc, loc2, vloc2, G, gin = _run_pair(g, torch.float32)
    msgs = check_parity(g, loc2, vloc2, G, gin, case_prefix="cpu_reassociation:")
    assert not msgs, msgs
