"""-m gpu: the device-side training step (fastegnn_amd/train.py: edge_attr augmentation, MSE+MMD loss,
backward through the HIP model, Adam) against the training-step goldens captured from the reference
(reference model + utils.train.kernel + torch.optim.Adam; oracle/gen_goldens.py --train-only)."""
import numpy as np
import pytest
import torch

import fastegnn_amd
from fastegnn_amd.train import FusedAdam, augment_edge_attr, mse_mmd_loss
from oracle import fastegnn_ref as R
from tests.helpers import grad_check, rel_err
from tests.test_train_oracle_cpu import NAMES, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", NAMES)
def test_three_training_steps_match_reference(name):
    g, cfg = load(name)
    m_ = g["meta"]
    model = fastegnn_amd.FastEGNN(2, 0, 2, 64, int(m_["C"]), device="cuda", n_layers=int(m_["L"]))
    model.load_state_dict(g["p0"], strict=True)
    model = model.cuda()
    opt = FusedAdam(model.parameters(), lr=float(m_["lr"]), weight_decay=float(m_["wd"]))
    inp = {k: v.cuda() for k, v in g["in"].items()}
    kw = {k: inp[k] for k in ("node_feat", "node_loc", "node_vel", "edge_index", "data_batch", "loc_mean", "edge_attr")}
    for step in range(1, 4):
        opt.zero_grad()
        loc, vloc = model(**kw)
        loss, mse = mse_mmd_loss(loc, vloc, inp["loc_t"], inp["sample_nodes"], float(m_["sigma"]), float(m_["weight"]))
        loss.backward()
        assert abs(loss.item() - float(g["out"]["losses"][step - 1, 0])) < 5e-6
        assert abs(float(mse) - float(g["out"]["losses"][step - 1, 1])) < 5e-6
        if step == 1:
            assert rel_err(loc, g["out"]["loc"]) < 1e-5 and rel_err(vloc, g["out"]["vloc"]) < 1e-5
            # fp64 truth of the step-1 gradients from the oracle; the reference's fp32 gradients are the goldens
            dt = torch.float64
            p64 = {k: v.to(dt).clone().requires_grad_(True) for k, v in g["p0"].items()}
            i64 = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in g["in"].items() if k not in ("loc_t", "sample_nodes")}
            l64, v64 = R.forward(p64, cfg, **i64)
            R.loss_mse_mmd_nodes(l64, v64, g["in"]["loc_t"].to(dt), g["in"]["sample_nodes"], float(m_["sigma"]),
                                 float(m_["weight"]))[0].backward()
            bad = []
            for k, p in model.named_parameters():
                got = p.grad if p.grad is not None else torch.zeros_like(p)
                tru = p64[k].grad if p64[k].grad is not None else torch.zeros_like(p64[k])
                grad_check(name, k, got, g["g1"][k], tru, bad)
            assert not bad, bad
        opt.step()
        if step in (1, 3):
            ref = g[f"p{step}"]
            sd = model.state_dict()
            bad = sum(int(((sd[k].cpu() - ref[k]).abs() > 5e-5).sum()) for k in ref)
            tot = sum(v.numel() for v in ref.values())
            # Adam normalises the gradient: entries whose gradient is pure rounding noise may flip sign
            assert bad <= 2e-4 * tot, (step, bad, tot)


def test_loss_gradient_matches_oracle_autograd():
    g = torch.Generator().manual_seed(4)
    N, B, Cn, S = 60, 3, 5, 7
    loc = torch.randn(N, 3, generator=g); tgt = torch.randn(N, 3, generator=g); vloc = torch.randn(B, 3, Cn, generator=g)
    vloc[0, :, 1] = vloc[0, :, 0]                         # coincident virtual nodes: zero-distance subgradient
    samp = torch.stack([b * 20 + torch.randperm(20, generator=g)[:S] for b in range(B)])
    a, v = loc.clone().requires_grad_(True), vloc.clone().requires_grad_(True)
    l_ref, mse_ref = R.loss_mse_mmd_nodes(a, v, tgt, samp, 1.3, 0.7)
    l_ref.backward()
    a2, v2 = loc.cuda().requires_grad_(True), vloc.cuda().requires_grad_(True)
    l, mse = mse_mmd_loss(a2, v2, tgt.cuda(), samp.cuda(), 1.3, 0.7)
    (2.0 * l).backward()
    assert abs(l.item() - l_ref.item()) < 1e-5 and abs(float(mse) - mse_ref.item()) < 1e-5
    assert rel_err(a2.grad, 2 * a.grad) < 1e-5 and rel_err(v2.grad, 2 * v.grad) < 1e-5


def test_augment_edge_attr():
    g = torch.Generator().manual_seed(1)
    loc = torch.randn(50, 3, generator=g); ei = torch.randint(0, 50, (2, 300), generator=g); ea = torch.rand(300, 1, generator=g)
    out = augment_edge_attr(ea.cuda(), loc.cuda(), ei.cuda()).cpu()
    assert torch.allclose(out, R.augment_edge_attr(ea, loc, ei), atol=1e-6)
    out0 = augment_edge_attr(None, loc.cuda(), ei.cuda()).cpu()
    assert torch.allclose(out0[:, 0], (loc[ei[0]] - loc[ei[1]]).norm(dim=1), atol=1e-6)
