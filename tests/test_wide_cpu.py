"""CPU: the orchestration of the wide path (fastegnn_amd/wide.py: hidden_nf > 64, EGNN flat=True) through a torch restatement of
the fastegnn_wide_* operators -- TEST INFRASTRUCTURE standing in for csrc/wide.hip, as tests/cpu_stage_backend.py does for
the sharded path -- against the goldens captured from the reference classes.  What this pins without a GPU: the op sequence, the
column-block bookkeeping of every Linear over a torch.cat, the (n, c) row layouts, the autograd wiring.  The operators themselves
are checked against the same torch arithmetic on the GPU (tests/test_gpu_wide.py)."""
import pytest
import torch
import torch.nn.functional as F

import fastegnn_amd
from fastegnn_amd import _lib as K
from fastegnn_amd import wide
from tests.helpers import Golden, check_parity, golden_loss, rel_err


def _fn(kind, p):
    return {K.ACT_SILU: F.silu, K.ACT_RELU: F.relu, K.ACT_LEAKY_RELU: lambda z: F.leaky_relu(z, p), K.ACT_TANH: torch.tanh,
            K.ACT_SIGMOID: torch.sigmoid, K.ACT_ELU: lambda z: F.elu(z, p), K.ACT_GELU: F.gelu,
            K.ACT_SOFTPLUS: lambda z: F.softplus(z, beta=p, threshold=20.0)}[kind]


class TorchOps:
    """include/fastegnn_hip.h "the WIDE path", one method per entry point, same argument order"""

    @staticmethod
    def call(name, *a):
        getattr(TorchOps, name)(*a)

    @staticmethod
    def _pro(X, kind, p):
        return X if kind == K.ACT_NONE else _fn(kind, p)(X)

    @staticmethod
    def linear(X, M, Kc, W, ldw, c0, bias, base, out, O, kind, p):
        assert W.shape == (O, ldw) and X.shape == (M, Kc) and out.shape == (M, O)
        r = TorchOps._pro(X, kind, p) @ W[:, c0:c0 + Kc].t()
        out.copy_(r + (bias if bias is not None else 0) + (base if base is not None else 0))

    @staticmethod
    def linear_dx(G, M, O, W, ldw, c0, Kc, dX, accumulate, Z, kind, p):
        r = G @ W[:, c0:c0 + Kc]
        if Z is not None:
            d = torch.empty_like(Z)
            TorchOps.act_backward(Z, r, Z.numel(), kind, p, d)
            r = d
        dX.copy_(dX + r if accumulate else r)

    @staticmethod
    def linear_dw(G, X, M, O, Kc, dW, ldw, c0, db, kind, p):
        if dW is not None:
            dW[:, c0:c0 + Kc] += G.t() @ TorchOps._pro(X, kind, p)
        if db is not None:
            db += G.sum(0)

    @staticmethod
    def _head_g(gs, w2, Zc, kind, p):
        d = torch.empty_like(Zc)
        TorchOps.act_backward(Zc, gs.reshape(-1, 1) * w2.reshape(1, -1), Zc.numel(), kind, p, d)
        return d

    @staticmethod
    def head_dx(gs, w2, Zc, M, O, W, ldw, c0, Kc, dX, accumulate, kind, p):
        TorchOps.linear_dx(TorchOps._head_g(gs, w2, Zc, kind, p), M, O, W, ldw, c0, Kc, dX, accumulate, None, K.ACT_NONE, 0.0)

    @staticmethod
    def head_dw(gs, w2, Zc, X, M, O, Kc, dW, ldw, c0, db, dw2, kind, p, x_kind, x_p):
        TorchOps.linear_dw(TorchOps._head_g(gs, w2, Zc, kind, p), X, M, O, Kc, dW, ldw, c0, db, x_kind, x_p)
        if dw2 is not None:
            dw2 += (gs.reshape(-1, 1) * _fn(kind, p)(Zc)).sum(0).reshape(dw2.shape)

    @staticmethod
    def head_forward(X, M, Kc, W1, ldw, c0, b1, w2, b2, Zc, s, O, kind, p, x_kind, x_p):
        TorchOps.linear(X, M, Kc, W1, ldw, c0, b1, None, Zc, O, x_kind, x_p)
        s.copy_(_fn(kind, p)(Zc) @ w2.reshape(-1, 1) + (b2 if b2 is not None else 0))

    @staticmethod
    def act(z, n, kind, p, y):
        y.copy_(_fn(kind, p)(z))

    @staticmethod
    def act_backward(z, dy, n, kind, p, dz):
        with torch.enable_grad():
            zz = z.detach().clone().requires_grad_(True)
            _fn(kind, p)(zz).backward(dy)
        dz.copy_(zz.grad)

    @staticmethod
    def gather_add(X, idx, M, W, base, out):
        out.copy_(X[idx] + (base if base is not None else 0))

    @staticmethod
    def gather2(P, i1, Q, i2, feat, nf, W, ldw, c0, base, out, M, Wd):
        r = P[i1]
        if Q is not None:
            r = r + Q[i2]
        if feat is not None and nf:
            r = r + feat @ W[:, c0:c0 + nf].t()
        out.copy_(r + (base if base is not None else 0))

    @staticmethod
    def scatter_add(table, idx, M, W, rows):
        table.index_add_(0, idx, rows)

    @staticmethod
    def act_scatter(z, idx, M, W, kind, p, y, table):
        y.copy_(_fn(kind, p)(z))
        table.index_add_(0, idx, y)

    @staticmethod
    def act_scatter_backward(z, idx, M, W, kind, p, g_y, g_table, dz):
        g = g_table[idx] + (g_y if g_y is not None else 0)
        TorchOps.act_backward(z, g, z.numel(), kind, p, dz)

    @staticmethod
    def scatter_add_perm(table, idx_sorted, perm, M, W, rows):
        table.index_add_(0, idx_sorted, rows[perm])

    @staticmethod
    def rowscale(X, s, M, W, Y):
        Y.copy_(X * s.unsqueeze(1))

    @staticmethod
    def rowdot(A, B, M, W, out):
        out.copy_((A * B).sum(1))


@pytest.fixture
def torch_ops(monkeypatch):
    monkeypatch.setattr(wide, "_OPS", TorchOps)


def test_wide_fastegnn_orchestration_matches_reference_golden(torch_ops):
    from tests.gpu_util import model_from_golden
    g = Golden("wide_h128_two_graphs")
    m = model_from_golden(g, device="cpu")
    kw, target, wv = g.model_kwargs()
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw.update(leaf)
    loc, vloc = wide.forward(m, **kw)
    golden_loss(loc, vloc, target, wv).backward()
    G = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    msgs = check_parity(g, loc.detach(), vloc.detach(), G, {k: v.grad for k, v in leaf.items()})
    assert not msgs, msgs


def test_wide_fastrf_orchestration_matches_reference_golden(torch_ops):
    from tests.gpu_util import model_from_golden
    g = Golden("fastrf_h128")
    m = model_from_golden(g, device="cpu", cls=fastegnn_amd.FastRF)
    kw, target, wv = g.model_kwargs()
    kw.pop("node_attr")
    loc, vloc = wide.forward(m, **kw)
    assert rel_err(loc, g.out["loc"]) < 1e-5 and rel_err(vloc, g.out["vloc"]) < 1e-5
    golden_loss(loc, vloc, target, wv).backward()
    for k, p in m.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert rel_err(got, g.gp[k]) < 5e-5, (k, rel_err(got, g.gp[k]))


@pytest.mark.parametrize("name", ["egnn_flat", "egnn_h128"])
def test_wide_egnn_orchestration_matches_reference_golden(torch_ops, name):
    from tests.test_egnn_oracle_cpu import egnn_loss, load_egnn
    g = load_egnn(name)
    m = fastegnn_amd.EGNN(n_layers=int(g["meta"]["L"]), in_node_nf=2, in_edge_nf=2, hidden_nf=int(g["meta"]["hidden"]),
                          with_v=bool(int(g["meta"]["with_v"])), norm=bool(int(g["meta"]["norm"])), flat=bool(int(g["meta"]["flat"])))
    m.load_state_dict(g["p"], strict=True)
    i = g["in"]
    x, h = wide.egnn_forward(m, i["x"], i["h"], i["edge_index"], i["edge_fea"], i.get("v"))
    assert rel_err(x, g["out"]["x"]) < 1e-5 and rel_err(h, g["out"]["h"]) < 2e-5
    egnn_loss(x, h, i["target"], i["wh"]).backward()
    for k, p in m.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert rel_err(got, g["gp"][k]) < 5e-5, (k, rel_err(got, g["gp"][k]))


@pytest.mark.parametrize("flags", [dict(tanh=True, normalize=True, gravity=[0.2, -1, 0.1]), dict(residual=False, attention=True),
                                   dict(act="elu", act_param=1.3, gravity=[0, -1, 0]), dict(coords_agg="sum")])
def test_wide_fastegnn_flags_match_the_oracle(torch_ops, flags):
    """every constructor flag (and coords_agg='sum', node_attr, a non-SiLU activation) through the wide orchestration against the
    oracle, fp32 torch arithmetic on both sides: only wiring differences would show"""
    from oracle import fastegnn_ref as R
    from tests.gpu_util import act_module
    hidden, C_, na = 72, 3, 2
    cfg = R.Config(2, na, 2, hidden, C_, n_layers=2, **flags)
    g = torch.Generator().manual_seed(3)
    sizes = [9, 6]
    N = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(2), torch.tensor(sizes))
    ei = torch.cat([torch.randint(0, 9, (2, 30), generator=g), 9 + torch.randint(0, 6, (2, 14), generator=g)], 1)
    loc = torch.randn(N, 3, generator=g)
    cm = torch.zeros(2, 3).index_add_(0, batch, loc) / torch.tensor(sizes).unsqueeze(1)
    inp = dict(node_feat=torch.rand(N, 2, generator=g), node_loc=loc, node_vel=torch.randn(N, 3, generator=g) * 0.3, edge_index=ei,
               data_batch=batch, loc_mean=cm.unsqueeze(-1).repeat(1, 1, C_), edge_attr=torch.rand(ei.size(1), 2, generator=g),
               node_attr=torch.rand(N, na, generator=g))
    p = R.init_params(cfg, seed=5, coord_gain=0.05)
    kw = {k: v for k, v in flags.items() if k not in ("act", "act_param", "coords_agg")}
    m = fastegnn_amd.FastEGNN(2, na, 2, hidden, C_, n_layers=2, act_fn=act_module(cfg), **kw)
    m.load_state_dict(p, strict=True)
    if flags.get("coords_agg") == "sum":
        m._extra_flags |= K.F_COORDS_SUM
    loc_w, vloc_w = wide.forward(m, **inp)
    (loc_w.pow(2).mean() + vloc_w.pow(2).mean()).backward()
    pp = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    l, v = R.forward(pp, cfg, **inp)
    (l.pow(2).mean() + v.pow(2).mean()).backward()
    assert rel_err(loc_w, l) < 2e-6 and rel_err(vloc_w, v) < 2e-6
    for k, prm in m.named_parameters():
        ref = pp[k].grad if pp[k].grad is not None else torch.zeros_like(pp[k])
        got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        assert rel_err(got, ref) < 2e-4, (k, rel_err(got, ref))
