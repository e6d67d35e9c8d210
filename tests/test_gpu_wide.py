"""-m gpu: the WIDE path (64 < hidden_nf <= 256; fastegnn_amd/wide.py on csrc/wide.hip).
(1) every fastegnn_wide_* operator against the same arithmetic in torch (test infrastructure) at ragged sizes,
(2) FastEGNN(hidden_nf = 128 / 96 / 160) forward and every gradient against the oracle (fp32 and fp64) with the repo's gradient
    rule, all constructor flags, (3) the golden captured from the reference at hidden_nf = 128 (tests/golden/h128_two_graphs.npz,
    oracle/gen_goldens.py)."""
import ctypes as C

import pytest
import torch

import fastegnn_amd
from fastegnn_amd import _lib as K
from fastegnn_amd import wide
from oracle import fastegnn_ref as R
from tests.helpers import rel_err
from tests.test_gpu_properties import _batch, _check_vs_oracle

pytestmark = pytest.mark.gpu


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("M,K_,O,ldw,c0", [(1000, 128, 128, 300, 37), (777, 3, 96, 200, 190), (513, 160, 1, 160, 0),
                                           (300, 2048, 128, 2304, 256), (65, 1, 130, 261, 256), (0, 16, 16, 16, 0),
                                           (5000, 96, 160, 256, 5), (33, 130, 256, 130, 0), (2100, 9, 24, 40, 3), (70000, 64, 128, 64, 0)])
@pytest.mark.parametrize("kind", [K.ACT_NONE, K.ACT_SILU, K.ACT_TANH])
def test_wide_linear_forward_dx_dw_vs_torch(M, K_, O, ldw, c0, kind):
    """the three Linear entry points at ragged shapes (every dispatch branch of csrc/wide.hip: the bf16x3 GEMM with 1-4 column
    quadrants and several k panels, the <= 8-column and <= 8-term kernels, the two small-side weight-gradient forms), plain and
    with the fused activation: act(X) in the forward's and the weight gradient's prologue, act'(Z) in the input gradient's epilogue"""
    if kind != K.ACT_NONE and M == 70000:
        pytest.skip("the large case once")
    g = torch.Generator().manual_seed(M + K_ + O)
    X = torch.randn(M, K_, generator=g).cuda()
    W = (torch.randn(O, ldw, generator=g) / max(K_, 1) ** 0.5).cuda()
    b = torch.randn(O, generator=g).cuda()
    base = torch.randn(M, O, generator=g).cuda()
    fn = (lambda t: t) if kind == K.ACT_NONE else {K.ACT_SILU: torch.nn.functional.silu, K.ACT_TANH: torch.tanh}[kind]
    L = K.lib()   # csrc/wide.hip carries every activation kind in every build
    out = torch.empty(M, O, device="cuda")
    K.check(L.fastegnn_wide_linear(K.ptr(X), M, K_, K.ptr(W), ldw, c0, K.ptr(b), K.ptr(base), K.ptr(out), O, kind, 0.0, _st()), "linear")
    Ws = W[:, c0:c0 + K_].double()
    Xa = fn(X.double())
    ref = base.double() + Xa @ Ws.t() + b.double()
    if M:
        assert rel_err(out.cpu(), ref.cpu()) < 2e-6
    G = torch.randn(M, O, generator=g).cuda()
    dX = torch.full((M, K_), 7.0, device="cuda")
    Z = None
    dref = G.double() @ Ws
    if kind != K.ACT_NONE:
        Z = torch.randn(M, K_, generator=g).cuda()
        zz = Z.double().requires_grad_(True)
        fn(zz).sum().backward()
        dref = dref * zz.grad
    K.check(L.fastegnn_wide_linear_dx(K.ptr(G), M, O, K.ptr(W), ldw, c0, K_, K.ptr(dX), 0, K.ptr(Z), kind, 0.0, _st()), "dx")
    if M:
        assert rel_err(dX.cpu(), dref.cpu()) < 2e-6
        K.check(L.fastegnn_wide_linear_dx(K.ptr(G), M, O, K.ptr(W), ldw, c0, K_, K.ptr(dX), 1, K.ptr(Z), kind, 0.0, _st()), "dx+")
        assert rel_err(dX.cpu(), (2 * dref).cpu()) < 2e-6
    dW = torch.ones(O, ldw, device="cuda")
    db = torch.ones(O, device="cuda")
    K.check(L.fastegnn_wide_linear_dw(K.ptr(G), K.ptr(X), M, O, K_, K.ptr(dW), ldw, c0, K.ptr(db), kind, 0.0, _st()), "dw")
    refW = torch.ones(O, ldw, dtype=torch.float64)
    refW[:, c0:c0 + K_] += (G.double().t() @ Xa).cpu()
    assert rel_err(dW.cpu(), refW) < 3e-6                      # columns outside the block untouched
    assert rel_err(db.cpu(), 1 + G.double().sum(0).cpu()) < 3e-6


@pytest.mark.parametrize("M,O,Kx", [(1000, 128, 128), (777, 96, 96), (4100, 256, 128), (500, 192, 224), (65, 160, 160)])
@pytest.mark.parametrize("kind", [K.ACT_SILU, K.ACT_TANH])
def test_wide_head_backward_vs_torch(M, O, Kx, kind):
    """fastegnn_wide_head_dx / _dw: the gradient of a scalar head's hidden pre-activation formed inside the GEMM kernels
    (gs[m] w2[o] act'(Zc[m, o])) against the same products of the materialised gradient in float64"""
    g = torch.Generator().manual_seed(M + O + Kx)
    fn = {K.ACT_SILU: torch.nn.functional.silu, K.ACT_TANH: torch.tanh}[kind]
    gs = torch.randn(M, generator=g).cuda()
    w2 = torch.randn(O, generator=g).cuda()
    Zc = torch.randn(M, O, generator=g).cuda()
    X = torch.randn(M, Kx, generator=g).cuda()
    ldw, c0 = Kx + 7, 3
    W1 = (torch.randn(O, ldw, generator=g) / Kx ** 0.5).cuda()
    zz = Zc.double().requires_grad_(True)
    fn(zz).sum().backward()
    G = gs.double().unsqueeze(1) * w2.double().unsqueeze(0) * zz.grad
    L = K.lib()
    dX = torch.full((M, Kx), 3.0, device="cuda")
    if O == 160:   # the generated-operand GEMM exists in the four-buffer form only: widths it does not take are refused, not run slowly
        assert L.fastegnn_wide_head_dx(K.ptr(gs), K.ptr(w2), K.ptr(Zc), M, O, K.ptr(W1), ldw, c0, Kx, K.ptr(dX), 0, kind, 0.0, _st()) != 0
        assert not wide._head_fits(X, W1[:, :Kx])
        return
    K.check(L.fastegnn_wide_head_dx(K.ptr(gs), K.ptr(w2), K.ptr(Zc), M, O, K.ptr(W1), ldw, c0, Kx, K.ptr(dX), 0, kind, 0.0, _st()), "head_dx")
    assert rel_err(dX.cpu(), (G @ W1[:, c0:c0 + Kx].double()).cpu()) < 2e-6
    dW = torch.ones(O, ldw, device="cuda")
    db = torch.ones(O, device="cuda")
    dw2 = torch.ones(O, device="cuda")
    K.check(L.fastegnn_wide_head_dw(K.ptr(gs), K.ptr(w2), K.ptr(Zc), K.ptr(X), M, O, Kx, K.ptr(dW), ldw, c0, K.ptr(db), K.ptr(dw2), kind, 0.0,
                                    K.ACT_NONE, 0.0, _st()), "head_dw")
    refW = torch.ones(O, ldw, dtype=torch.float64)
    refW[:, c0:c0 + Kx] += (G.t() @ X.double()).cpu()
    assert rel_err(dW.cpu(), refW) < 3e-6
    assert rel_err(db.cpu(), 1 + G.sum(0).cpu()) < 3e-6
    assert rel_err(dw2.cpu(), 1 + (gs.double().unsqueeze(1) * fn(Zc.double())).sum(0).cpu()) < 3e-6
    # the head's forward: the hidden pre-activation and the scalar output (from the first GEMM's accumulators when O <= 128)
    b1 = torch.randn(O, generator=g).cuda()
    b2 = torch.randn(1, generator=g).cuda()
    zc = torch.empty(M, O, device="cuda")
    so = torch.empty(M, 1, device="cuda")
    K.check(L.fastegnn_wide_head_forward(K.ptr(X), M, Kx, K.ptr(W1), ldw, c0, K.ptr(b1), K.ptr(w2), K.ptr(b2), K.ptr(zc), K.ptr(so), O, kind, 0.0,
                                         K.ACT_NONE, 0.0, _st()), "head_forward")
    zr = X.double() @ W1[:, c0:c0 + Kx].double().t() + b1.double()
    assert rel_err(zc.cpu(), zr.cpu()) < 2e-6
    assert rel_err(so.cpu(), (fn(zr) @ w2.double().unsqueeze(1) + b2.double()).cpu()) < 2e-6


def test_wide_rowwise_operators_vs_torch():
    g = torch.Generator().manual_seed(5)
    L = K.lib()
    R_, M, W = 400, 3001, 136
    X = torch.randn(R_, W, generator=g).cuda()
    idx = torch.randint(0, R_, (M,), generator=g).cuda()
    base = torch.randn(M, W, generator=g).cuda()
    out = torch.empty(M, W, device="cuda")
    K.check(L.fastegnn_wide_gather_add(K.ptr(X), K.ptr(idx), M, W, K.ptr(base), K.ptr(out), _st()), "gather")
    assert torch.equal(out, base + X[idx])
    K.check(L.fastegnn_wide_gather_add(K.ptr(X), K.ptr(idx), M, W, None, K.ptr(out), _st()), "gather")
    assert torch.equal(out, X[idx])
    table = torch.zeros(R_, W, device="cuda")
    K.check(L.fastegnn_wide_scatter_add(K.ptr(table), K.ptr(idx), M, W, K.ptr(base), _st()), "scatter")
    ref = torch.zeros(R_, W, dtype=torch.float64).index_add_(0, idx.cpu(), base.double().cpu())
    assert rel_err(table.cpu(), ref) < 1e-6
    # sorted targets (runs), the same through a sorting permutation, and rows wider than one workgroup's columns
    sidx, perm = torch.sort(idx, stable=True)
    for tgt, pm in ((sidx, None), (sidx, perm)):
        table = torch.zeros(R_, W, device="cuda")
        if pm is None:
            K.check(L.fastegnn_wide_scatter_add(K.ptr(table), K.ptr(tgt), M, W, K.ptr(base), _st()), "scatter sorted")
            ref = torch.zeros(R_, W, dtype=torch.float64).index_add_(0, tgt.cpu(), base.double().cpu())
        else:
            K.check(L.fastegnn_wide_scatter_add_perm(K.ptr(table), K.ptr(tgt), K.ptr(pm), M, W, K.ptr(base), _st()), "scatter perm")
            ref = torch.zeros(R_, W, dtype=torch.float64).index_add_(0, idx.cpu(), base.double().cpu())
        assert rel_err(table.cpu(), ref) < 1e-6
    for Wd, nf in ((128, 3), (2048, 0), (96, 1), (30, 2), (1300, 8)):
        P = torch.randn(R_, Wd, generator=g).cuda()
        Q = torch.randn(77, Wd, generator=g).cuda()
        i2 = torch.randint(0, 77, (M,), generator=g).cuda()
        feat = torch.randn(M, max(nf, 1), generator=g).cuda()
        Wf = torch.randn(Wd, 40, generator=g).cuda()
        bs = torch.randn(M, Wd, generator=g).cuda()
        o = torch.empty(M, Wd, device="cuda")
        K.check(L.fastegnn_wide_gather2(K.ptr(P), K.ptr(idx), K.ptr(Q), K.ptr(i2), K.ptr(feat) if nf else None, nf, K.ptr(Wf), 40, 5,
                                        K.ptr(bs), K.ptr(o), M, Wd, _st()), "gather2")
        ref = bs.double() + P.double()[idx] + Q.double()[i2]
        if nf:
            ref = ref + feat.double() @ Wf[:, 5:5 + nf].double().t()
        assert rel_err(o.cpu(), ref.cpu()) < 1e-6, (Wd, nf)
        K.check(L.fastegnn_wide_gather2(K.ptr(P), K.ptr(idx), None, None, None, 0, None, 0, 0, None, K.ptr(o), M, Wd, _st()), "gather2 plain")
        assert torch.equal(o, P[idx])
        t2 = torch.zeros(R_, Wd, device="cuda")
        K.check(L.fastegnn_wide_scatter_add_perm(K.ptr(t2), K.ptr(sidx), K.ptr(perm), M, Wd, K.ptr(bs), _st()), "scatter perm wide") if Wd >= 32 else \
            K.check(L.fastegnn_wide_scatter_add(K.ptr(t2), K.ptr(idx), M, Wd, K.ptr(bs), _st()), "scatter narrow")
        ref = torch.zeros(R_, Wd, dtype=torch.float64).index_add_(0, idx.cpu(), bs.double().cpu())
        assert rel_err(t2.cpu(), ref) < 1e-6, Wd
    # the activation and the segment sum of its output in one pass, and the pair's backward
    for Wd in (128, 2048, 20):
        zz = torch.randn(M, Wd, generator=g).cuda()
        y = torch.empty_like(zz)
        tb = torch.zeros(R_, Wd, device="cuda")
        K.check(L.fastegnn_wide_act_scatter(K.ptr(zz), K.ptr(sidx), M, Wd, K.ACT_SILU, 0.0, K.ptr(y), K.ptr(tb), _st()), "act_scatter")
        yr = torch.nn.functional.silu(zz.double())
        assert rel_err(y.cpu(), yr.cpu()) < 1e-6
        assert rel_err(tb.cpu(), torch.zeros(R_, Wd, dtype=torch.float64).index_add_(0, sidx.cpu(), yr.cpu())) < 1e-6
        gy, gt = torch.randn(M, Wd, generator=g).cuda(), torch.randn(R_, Wd, generator=g).cuda()
        dz = torch.empty_like(zz)
        K.check(L.fastegnn_wide_act_scatter_backward(K.ptr(zz), K.ptr(sidx), M, Wd, K.ACT_SILU, 0.0, K.ptr(gy), K.ptr(gt), K.ptr(dz), _st()), "asb")
        zd = zz.double().requires_grad_(True)
        torch.nn.functional.silu(zd).backward(gy.double() + gt.double()[sidx])
        assert rel_err(dz.cpu(), zd.grad.cpu()) < 2e-6
    s = torch.randn(M, generator=g).cuda()
    Y = torch.empty(M, W, device="cuda")
    K.check(L.fastegnn_wide_rowscale(K.ptr(base), K.ptr(s), M, W, K.ptr(Y), _st()), "rowscale")
    assert torch.equal(Y, base * s.unsqueeze(1))
    d = torch.empty(M, device="cuda")
    K.check(L.fastegnn_wide_rowdot(K.ptr(base), K.ptr(Y), M, W, K.ptr(d), _st()), "rowdot")
    assert rel_err(d.cpu(), (base.double() * Y.double()).sum(1).cpu()) < 2e-6
    # activations: value and derivative of every kind against autograd
    z = (torch.randn(5000, generator=g) * 3).cuda()
    mods = {K.ACT_SILU: torch.nn.SiLU(), K.ACT_RELU: torch.nn.ReLU(), K.ACT_LEAKY_RELU: torch.nn.LeakyReLU(0.2), K.ACT_TANH: torch.nn.Tanh(),
            K.ACT_SIGMOID: torch.nn.Sigmoid(), K.ACT_ELU: torch.nn.ELU(1.3), K.ACT_GELU: torch.nn.GELU(), K.ACT_SOFTPLUS: torch.nn.Softplus(beta=1.7)}
    par = {K.ACT_LEAKY_RELU: 0.2, K.ACT_ELU: 1.3, K.ACT_SOFTPLUS: 1.7}
    for kind, mod in mods.items():
        zz = z.double().cpu().requires_grad_(True)
        yy = mod(zz)
        yy.sum().backward()
        y, dz = torch.empty_like(z), torch.empty_like(z)
        K.check(L.fastegnn_wide_act(K.ptr(z), z.numel(), kind, par.get(kind, 0.0), K.ptr(y), _st()), "act")
        K.check(L.fastegnn_wide_act_backward(K.ptr(z), K.ptr(torch.ones_like(z)), z.numel(), kind, par.get(kind, 0.0), K.ptr(dz), _st()), "dact")
        assert rel_err(y.cpu(), yy.detach()) < 2e-6, kind
        assert rel_err(dz.cpu(), zz.grad) < 5e-6, kind


@pytest.mark.parametrize("hidden,flags", [(128, dict(gravity=[0, -1, 0])),
                                          (128, dict(attention=True, tanh=True, gravity=[0.3, -1, 0.2])),
                                          (96, dict(residual=False)),
                                          (160, dict(attention=True, act="gelu")),
                                          (256, dict(gravity=[0, -1, 0])),          # two column blocks, two contraction panels
                                          (192, dict(act="tanh", residual=False)),  # the four-buffer GEMM on padded panels, generic head forms
                                          (224, dict())])
def test_wide_model_vs_oracle(hidden, flags):
    """FastEGNN(hidden_nf > 64): outputs <= 1e-5 of the fp32 oracle, displacement and every parameter gradient within the
    repo's rule (2 x the fp32 reference's own error against fp64 + 1e-6, tests/helpers.py)."""
    C_ = 3
    cfg = R.Config(2, 0, 2, hidden, C_, n_layers=2, **flags)
    _check_vs_oracle(cfg, _batch([300, 141, 77], 6, C_, seed=hidden), seed=hidden, case=f"wide_h{hidden}")


@pytest.mark.parametrize("nf,ea,C_", [(9, 2, 3), (2, 9, 3), (2, 2, 70)])
def test_beyond_the_fused_kernels_argument_ceilings_vs_oracle(nf, ea, C_):
    """hidden_nf = 64 with node_feat_nf > 8, edge_attr_nf > 7 or virtual_channels > 64: the reference has no such limits
    (models/FastEGNN.py:227-263); the module routes these shapes to the wide path instead of refusing them."""
    cfg = R.Config(nf, 0, ea, 64, C_, n_layers=2, gravity=[0, -1, 0])
    assert fastegnn_amd.FastEGNN(nf, 0, ea, 64, C_)._wide
    _check_vs_oracle(cfg, _batch([150, 60], 5, C_, seed=nf + ea + C_, nf=nf, ea=ea), seed=nf + ea + C_, case="wide_ceilings")


def test_wide_model_normalize_vs_oracle():
    """normalize=True (:181-183) on a graph WITHOUT self loops: d / (|d| + 1e-8) at d = 0 amplifies rounding noise by 1e8 (the
    reference's own fp32 gradients sit 1e-3 from exact arithmetic on such edges -- tests/helpers.py, the ragged3_normalize entry),
    which says nothing about a kernel; away from d = 0 the plain rule applies."""
    hidden, C_ = 128, 3
    cfg = R.Config(2, 0, 2, hidden, C_, n_layers=2, normalize=True, attention=True, gravity=[0, -1, 0])
    inp = _batch([300, 141, 77], 6, C_, seed=77)
    keep = inp["edge_index"][0] != inp["edge_index"][1]
    inp["edge_index"], inp["edge_attr"] = inp["edge_index"][:, keep].contiguous(), inp["edge_attr"][keep].contiguous()
    _check_vs_oracle(cfg, inp, seed=77, case="wide_h128_normalize")


def test_wide_model_node_attr_and_input_gradients():
    """node_attr / edge_attr widths outside the fused path's limits are fine here; gradients w.r.t. every floating-point input"""
    hidden, C_, na, ea = 128, 4, 3, 9
    cfg = R.Config(2, na, ea, hidden, C_, n_layers=2, gravity=[0, -1, 0])
    inp = _batch([120, 60], 5, C_, seed=9, ea=ea)
    g = torch.Generator().manual_seed(1)
    inp["node_attr"] = torch.rand(inp["node_loc"].size(0), na, generator=g)
    p = R.init_params(cfg, seed=3, coord_gain=0.05)
    m = fastegnn_amd.FastEGNN(2, na, ea, hidden, C_, device="cuda", n_layers=2, gravity=[0, -1, 0])
    m.load_state_dict(p, strict=True)
    leaves = {k: inp[k].clone().cuda().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean", "edge_attr", "node_attr")}
    loc, vloc = m(edge_index=inp["edge_index"].cuda(), data_batch=inp["data_batch"].cuda(), **leaves)
    (loc.pow(2).mean() + vloc.pow(2).mean()).backward()
    dt = torch.float64
    pp = {k: v.detach().to(dt) for k, v in p.items()}
    l64 = {k: inp[k].to(dt).clone().requires_grad_(True) for k in leaves}
    l, v = R.forward(pp, cfg, edge_index=inp["edge_index"], data_batch=inp["data_batch"], **l64)
    (l.pow(2).mean() + v.pow(2).mean()).backward()
    assert rel_err(loc, l) < 1e-5 and rel_err(vloc, v) < 1e-5
    for k in leaves:
        assert rel_err(leaves[k].grad.cpu(), l64[k].grad) < 2e-4, (k, rel_err(leaves[k].grad.cpu(), l64[k].grad))


def test_wide_model_matches_reference_golden():
    """hidden_nf = 128 captured from the REAL reference (oracle/gen_goldens.py --wide): outputs, per-layer states' end result and
    every parameter / input gradient under the parity suite's own rule (tests/helpers.py check_parity)"""
    from tests.gpu_util import model_from_golden
    from tests.helpers import Golden, check_parity, golden_loss
    g = Golden("wide_h128_two_graphs")
    m = model_from_golden(g)
    assert m._wide
    kw, target, wv = g.model_kwargs(device="cuda")
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw.update(leaf)
    loc, vloc = m(**kw)
    golden_loss(loc, vloc, target, wv).backward()
    G = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    msgs = check_parity(g, loc.detach(), vloc.detach(), G, {k: v.grad for k, v in leaf.items()})
    assert not msgs, msgs


def test_wide_model_without_edges():
    """E = 0 (every node isolated): the edge operators see empty inputs, segment means of nothing are zero (count.clamp(min=1))"""
    hidden, C_ = 96, 2
    cfg = R.Config(2, 0, 2, hidden, C_, n_layers=2, gravity=[0, -1, 0])
    inp = _batch([9, 5], 3, C_, seed=4)
    inp["edge_index"] = torch.zeros(2, 0, dtype=torch.long)
    inp["edge_attr"] = torch.zeros(0, 2)
    _check_vs_oracle(cfg, inp, seed=4, case="wide_no_edges")
