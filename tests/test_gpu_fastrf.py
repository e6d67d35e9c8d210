"""-m gpu: the HIP-backed FastRF module (FASTEGNN_F_RF on the FastEGNN stage kernels) against goldens captured
from the reference's FastRF class and against the fp64 oracle."""
import pytest
import torch

import fastegnn_amd
from oracle import fastrf_ref as RF
from tests.gpu_util import model_from_golden
from tests.helpers import Golden, golden_loss, grad_check, rel_err, OUT_TOL

pytestmark = pytest.mark.gpu
CASES = ["fastrf_plain", "fastrf_allflags", "fastrf_c16", "fastrf_h128"]   # fastrf_h128: hidden_nf = 128, the unfused wide path


def _truth(g):
    dt = torch.float64
    p = {k: v.clone().requires_grad_(True) for k, v in g.tensors(g.params, dtype=dt).items()}
    kw, target, wv = g.model_kwargs(dtype=dt)
    kw.pop("node_attr")
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw.update(leaf)
    loc, vloc = RF.forward(p, g.cfg, **kw)
    golden_loss(loc, vloc, target, wv).backward()
    G = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    gin = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return G, gin


@pytest.mark.parametrize("name", CASES)
def test_fastrf_forward_backward_match_reference_golden(name):
    g = Golden(name)
    m = model_from_golden(g, cls=fastegnn_amd.FastRF)
    kw, target, wv = g.model_kwargs(device="cuda")
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw.update(leaf)
    loc, vloc = m(**kw)
    assert rel_err(loc, g.out["loc"]) < OUT_TOL and rel_err(vloc, g.out["vloc"]) < OUT_TOL   # 1e-5 rel (north_star)
    golden_loss(loc, vloc, target, wv).backward()
    tG, tgin = _truth(g)
    bad = []
    for k, p in m.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        grad_check(name, f"gp/{k}", got, g.gp[k], tG[k], bad)
    for k, v in leaf.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        grad_check(name, f"gin/{k}", got, g.gin[k], tgin[k], bad)
    assert not bad, bad


def test_fastrf_features_pass_through_and_velocity_scale_ignores_h():
    """Zero velocity removes the velocity term whatever the weights are; the same weights in FastEGNN's velocity
    head would not (it reads h): guards against wiring the wrong head."""
    g = Golden("fastrf_plain")
    m = model_from_golden(g, cls=fastegnn_amd.FastRF)
    kw, _, _ = g.model_kwargs(device="cuda")
    with torch.no_grad():
        a, _ = m(**kw)
        kw2 = dict(kw); kw2["node_vel"] = torch.zeros_like(kw["node_vel"])
        b, _ = m(**kw2)
        p = {k: v.cpu() for k, v in m.state_dict().items()}
        kwc = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in kw2.items()}; kwc.pop("node_attr")
        ref, _ = RF.forward(p, g.cfg, **kwc)
    assert rel_err(b, ref) < OUT_TOL and rel_err(a, b) > 1e-4


def test_fastrf_many_tiles_per_workgroup_vs_oracle():
    """Large enough that the virtual kernels take their channel-split last-tile path under FASTEGNN_F_RF too."""
    from oracle import fastegnn_ref as R
    from tests.test_gpu_properties import _batch
    cfg = R.Config(2, 0, 2, 64, 8, n_layers=2, gravity=[0, -1, 0])
    inp = _batch([20000, 16900], 2, 8, seed=21)
    m = fastegnn_amd.FastRF(2, 0, 2, 64, 8, device="cuda", n_layers=2, gravity=[0, -1, 0])
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(("coord_mlp_r.2.weight", "coord_mlp_r_virtual.2.weight", "coord_mlp_v_virtual.2.weight")):
                p.mul_(50.0)
    p32 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    tgt = inp["node_loc"] + 0.5
    loc, vloc = m(**{k: v.cuda() for k, v in inp.items()})
    (torch.nn.functional.mse_loss(loc, tgt.cuda()) + 0.05 * vloc.pow(2).mean()).backward()
    res = {}
    for dt in (torch.float32, torch.float64):
        pp = {k: v.to(dt).clone().requires_grad_(True) for k, v in p32.items()}
        ii = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in inp.items()}
        l, v = RF.forward(pp, cfg, **ii)
        (torch.nn.functional.mse_loss(l, tgt.to(dt)) + 0.05 * v.pow(2).mean()).backward()
        res[dt] = (l.detach(), v.detach(), {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in pp.items()})
    l32, v32, g32 = res[torch.float32]
    _, _, g64 = res[torch.float64]
    assert rel_err(loc, l32) < OUT_TOL and rel_err(vloc, v32) < OUT_TOL
    bad = []
    for k, p in m.named_parameters():
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(p32[k])
        grad_check("fastrf_many_tiles", k, got, g32[k], g64[k], bad)
    assert not bad, bad


def test_fastrf_narrow_hidden_nf_vs_oracle():
    """hidden_nf = 24 on the FastRF sibling: zero-padded parameters on the 64-wide kernels (model.py:_pad_param, rf=True)
    against the oracle evaluated with the true hidden_nf: outputs and parameter gradients."""
    from oracle import fastegnn_ref as R
    from tests.test_gpu_properties import _batch
    h, C = 24, 3
    cfg = R.Config(2, 0, 2, h, C, n_layers=2, gravity=[0, -1, 0])
    inp = _batch([90, 41], 5, C, seed=31)
    torch.manual_seed(9)
    m = fastegnn_amd.FastRF(2, 0, 2, h, C, device="cuda", n_layers=2, gravity=[0, -1, 0])
    tgt = inp["node_loc"] + 0.5
    loc, vloc = m(**{k: v.cuda() for k, v in inp.items()})
    (torch.nn.functional.mse_loss(loc, tgt.cuda()) + 0.05 * vloc.pow(2).mean()).backward()
    res = {}
    for dt in (torch.float32, torch.float64):
        p = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        ii = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in inp.items()}
        l, v = RF.forward(p, cfg, **ii)
        (torch.nn.functional.mse_loss(l, tgt.to(dt)) + 0.05 * v.pow(2).mean()).backward()
        res[dt] = (l.detach(), v.detach(), {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in p.items()})
    l32, v32, g32 = res[torch.float32]
    _, _, g64 = res[torch.float64]
    assert rel_err(loc, l32) < OUT_TOL and rel_err(vloc, v32) < OUT_TOL
    bad = []
    for k, prm in m.named_parameters():
        got = prm.grad.cpu() if prm.grad is not None else torch.zeros_like(prm).cpu()
        assert got.shape == g64[k].shape
        grad_check("test_fastrf_narrow_hidden_nf_vs_oracle", k, got, g32[k], g64[k], bad)
    assert not bad, bad


@pytest.mark.parametrize("act,q,h", [("elu", 1.0, 64), ("tanh", 0.0, 24)])
def test_fastrf_other_activation_vs_oracle(act, q, h):
    """act_fn other than SiLU on the FastRF sibling (generic-activation library, FASTEGNN_F_RF wiring incl. its velocity head
    on the 1-wide speed input), once together with zero-padded narrow parameters (Tanh(0) = 0)."""
    from oracle import fastegnn_ref as R
    from tests.gpu_util import act_module
    from tests.test_gpu_properties import _batch
    C = 3
    cfg = R.Config(2, 0, 2, h, C, n_layers=2, gravity=[0, -1, 0], attention=True, act=act, act_param=q)
    inp = _batch([300, 141], 5, C, seed=33)
    torch.manual_seed(10)
    m = fastegnn_amd.FastRF(2, 0, 2, h, C, device="cuda", n_layers=2, gravity=[0, -1, 0], attention=True, act_fn=act_module(cfg))
    tgt = inp["node_loc"] + 0.5
    loc, vloc = m(**{k: v.cuda() for k, v in inp.items()})
    (torch.nn.functional.mse_loss(loc, tgt.cuda()) + 0.05 * vloc.pow(2).mean()).backward()
    res = {}
    for dt in (torch.float32, torch.float64):
        p = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        ii = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in inp.items()}
        l, v = RF.forward(p, cfg, **ii)
        (torch.nn.functional.mse_loss(l, tgt.to(dt)) + 0.05 * v.pow(2).mean()).backward()
        res[dt] = (l.detach(), v.detach(), {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in p.items()})
    l32, v32, g32 = res[torch.float32]
    _, _, g64 = res[torch.float64]
    assert rel_err(loc, l32) < OUT_TOL and rel_err(vloc, v32) < OUT_TOL
    bad = []
    for k, prm in m.named_parameters():
        got = prm.grad.cpu() if prm.grad is not None else torch.zeros_like(prm).cpu()
        grad_check(f"act_mid_fastrf_{act}_attention", k, got, g32[k], g64[k], bad)
    assert not bad, bad
