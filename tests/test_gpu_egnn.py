"""-m gpu: the EGNN baseline on the HIP kernels (fastegnn_amd/egnn.py, FASTEGNN_F_EGNN wiring) against goldens
captured from the reference's models/basic.py EGNN: outputs and every gradient; plus a mid-size oracle check."""
import pytest
import torch

import fastegnn_amd
from oracle import egnn_ref as E
from tests.helpers import grad_check, rel_err
from tests.test_egnn_oracle_cpu import EGNN_NAMES, EGNN_WIDE_NAMES, egnn_act, egnn_loss, load_egnn

pytestmark = pytest.mark.gpu


# (EGNN_WIDE_NAMES: flat=True and hidden_nf = 128 -- the sibling on the unfused wide path, fastegnn_amd/wide.py)
@pytest.mark.parametrize("name", EGNN_NAMES + EGNN_WIDE_NAMES)
def test_egnn_matches_reference_golden(name):
    g = load_egnn(name)
    with_v = bool(int(g["meta"]["with_v"]))
    norm = bool(int(g["meta"].get("norm", 0)))
    hidden, flat = int(g["meta"].get("hidden", 64)), bool(int(g["meta"].get("flat", 0)))
    m = fastegnn_amd.EGNN(n_layers=int(g["meta"]["L"]), in_node_nf=2, in_edge_nf=2, hidden_nf=hidden, device="cuda", with_v=with_v,
                          norm=norm, flat=flat)
    assert m._wide == (name in EGNN_WIDE_NAMES)
    assert list(m.state_dict().keys()) == list(g["p"].keys())
    m.load_state_dict(g["p"], strict=True)
    m = m.cuda()
    i = {k: v.cuda() for k, v in g["in"].items()}
    leaf = {k: i[k].clone().requires_grad_(True) for k in ("x", "h") + (("v",) if with_v else ())}
    out = m(x=leaf["x"], h=leaf["h"], edge_index=i["edge_index"], edge_fea=i["edge_fea"], v=leaf.get("v"))
    x, h = out[0], out[-1]
    assert len(out) == (3 if with_v else 2)
    assert rel_err(x, g["out"]["x"]) < 1e-5, rel_err(x, g["out"]["x"])
    assert rel_err(h, g["out"]["h"]) < 2e-5
    egnn_loss(x, h, i["target"], i["wh"]).backward()
    # fp64 truth from the oracle on the golden's fp32 weights / inputs; the reference's fp32 gradients are the goldens
    dt = torch.float64
    p64 = {k: v.to(dt).clone().requires_grad_(True) for k, v in g["p"].items()}
    l64 = {k: g["in"][k].to(dt).clone().requires_grad_(True) for k in leaf}
    x64, h64 = E.forward(p64, int(g["meta"]["L"]), l64["x"], l64["h"], g["in"]["edge_index"], g["in"]["edge_fea"].to(dt), l64.get("v"), norm=norm,
                         act=egnn_act(g))
    egnn_loss(x64, h64, g["in"]["target"].to(dt), g["in"]["wh"].to(dt)).backward()
    bad = []
    for k, p in m.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        tru = p64[k].grad if p64[k].grad is not None else torch.zeros_like(p64[k])
        grad_check(name, f"gp/{k}", got, g["gp"][k], tru, bad)
    for k, v in leaf.items():
        grad_check(name, f"gin/{k}", v.grad, g["gin"][k], l64[k].grad, bad)
    assert not bad, bad


def test_egnn_mid_size_vs_oracle():
    g = torch.Generator().manual_seed(3)
    N, Ed = 3000, 40000
    torch.manual_seed(5)
    m = fastegnn_amd.EGNN(n_layers=3, in_node_nf=2, in_edge_nf=2, hidden_nf=64, device="cuda", with_v=True)
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.named_parameters()}
    x = torch.randn(N, 3, generator=g); h = torch.rand(N, 2, generator=g); v = torch.randn(N, 3, generator=g) * 0.2
    ei = torch.randint(0, N, (2, Ed), generator=g); ea = torch.rand(Ed, 2, generator=g)
    # edge_fea is a differentiable input of the reference layer (basic.py:313): its gradient comes back in the caller's edge order
    ea_gpu = ea.cuda().requires_grad_(True)
    xo, vo, ho = m(x=x.cuda(), h=h.cuda(), edge_index=ei.cuda(), edge_fea=ea_gpu, v=v.cuda())
    (xo.pow(2).mean() + ho.pow(2).mean()).backward()
    ea32 = ea.clone().requires_grad_(True)
    xr, hr = E.forward(p, 3, x, h, ei, ea32, v)
    (xr.pow(2).mean() + hr.pow(2).mean()).backward()
    assert rel_err(xo, xr) < 1e-5 and rel_err(ho, hr) < 2e-5
    dt = torch.float64
    p64 = {k: t.detach().to(dt).clone().requires_grad_(True) for k, t in p.items()}
    ea64 = ea.to(dt).requires_grad_(True)
    x64, h64 = E.forward(p64, 3, x.to(dt), h.to(dt), ei, ea64, v.to(dt))
    (x64.pow(2).mean() + h64.pow(2).mean()).backward()
    bad = []
    for k, prm in m.named_parameters():
        grad_check("egnn_mid_size", k, prm.grad, p[k].grad, p64[k].grad, bad)
    assert ea_gpu.grad is not None and ea_gpu.grad.shape == ea.shape
    grad_check("egnn_mid_size", "gin/edge_fea", ea_gpu.grad.cpu(), ea32.grad, ea64.grad, bad)
    assert not bad, bad


@pytest.mark.parametrize("hidden", [24, 40])
def test_egnn_narrow_hidden_nf_vs_oracle(hidden):
    """--dim_hidden below 64 on the EGNN sibling (main_nbody.py:107): the parameters run zero-padded on the 64-wide tiles
    (fastegnn_pad_params with the EGNN block layout: one leading radial column in the message MLP's input), the returned h
    has the reference's width, gradients arrive in the reference's shapes."""
    g = torch.Generator().manual_seed(11)
    N, Ed = 900, 9000
    torch.manual_seed(6)
    m = fastegnn_amd.EGNN(n_layers=2, in_node_nf=2, in_edge_nf=2, hidden_nf=hidden, device="cuda", with_v=True)
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.named_parameters()}
    x = torch.randn(N, 3, generator=g); h = torch.rand(N, 2, generator=g); v = torch.randn(N, 3, generator=g) * 0.2
    ei = torch.randint(0, N, (2, Ed), generator=g); ea = torch.rand(Ed, 2, generator=g)
    xo, vo, ho = m(x=x.cuda(), h=h.cuda(), edge_index=ei.cuda(), edge_fea=ea.cuda(), v=v.cuda())
    assert ho.shape == (N, hidden)
    (xo.pow(2).mean() + ho.pow(2).mean()).backward()
    xr, hr = E.forward(p, 2, x, h, ei, ea, v)
    (xr.pow(2).mean() + hr.pow(2).mean()).backward()
    assert rel_err(xo, xr) < 1e-5 and rel_err(ho, hr) < 2e-5
    dt = torch.float64
    p64 = {k: t.detach().to(dt).clone().requires_grad_(True) for k, t in p.items()}
    x64, h64 = E.forward(p64, 2, x.to(dt), h.to(dt), ei, ea.to(dt), v.to(dt))
    (x64.pow(2).mean() + h64.pow(2).mean()).backward()
    bad = []
    for k, prm in m.named_parameters():
        assert prm.grad is not None and prm.grad.shape == prm.shape, k
        grad_check("egnn_narrow", k, prm.grad, p[k].grad, p64[k].grad, bad)
    assert not bad, bad


@pytest.mark.parametrize("act,q", [("relu", 0.0), ("leaky_relu", 0.2), ("gelu", 0.0), ("tanh", 0.0)])
def test_egnn_other_activation_vs_oracle(act, q):
    """`activation` other than SiLU on the EGNN sibling (basic.py:324; BaseMLP :181-192 puts it behind every first layer and
    behind the message MLP): the generic-activation library on the FASTEGNN_F_EGNN wiring, forward and every gradient."""
    from oracle import fastegnn_ref as R
    from tests.gpu_util import act_module
    cfg = R.Config(2, 0, 2, 64, 0, act=act, act_param=q)
    fn = R.act_of(cfg)
    g = torch.Generator().manual_seed(17)
    N, Ed = 1500, 18000
    torch.manual_seed(8)
    m = fastegnn_amd.EGNN(n_layers=2, in_node_nf=2, in_edge_nf=2, hidden_nf=64, activation=act_module(cfg), device="cuda", with_v=True)
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.named_parameters()}
    x = torch.randn(N, 3, generator=g); h = torch.rand(N, 2, generator=g); v = torch.randn(N, 3, generator=g) * 0.2
    ei = torch.randint(0, N, (2, Ed), generator=g); ea = torch.rand(Ed, 2, generator=g)
    xo, _, ho = m(x=x.cuda(), h=h.cuda(), edge_index=ei.cuda(), edge_fea=ea.cuda(), v=v.cuda())
    (xo.pow(2).mean() + ho.pow(2).mean()).backward()
    xr, hr = E.forward(p, 2, x, h, ei, ea, v, act=fn)
    (xr.pow(2).mean() + hr.pow(2).mean()).backward()
    assert rel_err(xo, xr) < 1e-5 and rel_err(ho, hr) < 2e-5
    dt = torch.float64
    p64 = {k: t.detach().to(dt).clone().requires_grad_(True) for k, t in p.items()}
    x64, h64 = E.forward(p64, 2, x.to(dt), h.to(dt), ei, ea.to(dt), v.to(dt), act=fn)
    (x64.pow(2).mean() + h64.pow(2).mean()).backward()
    bad = []
    for k, prm in m.named_parameters():
        grad_check(f"egnn_act_{act}", k, prm.grad, p[k].grad, p64[k].grad, bad)
    assert not bad, bad
