"""The WIDE path: ``FastEGNN`` with ``64 < hidden_nf <= 256`` (the reference takes any ``--dim_hidden``:
``main_nbody.py:27``, ``models/FastEGNN.py:28-99``), the shapes beyond the fused kernels' argument ceilings, ``EGNN(flat=True)``.

The fused stage kernels are built on 64-wide register tiles; a wider model runs the op sequence of
``models/FastEGNN.py:102-223`` (edge_model, edge_mode_virtual, coord_model_vel, coord_model_virtual, node_model,
node_model_virtual) on the operators of ``csrc/wide.hip`` + ``csrc/wide_gemm.h`` behind the C ABI (``fastegnn_wide_*`` in
``include/fastegnn_hip.h``), wrapped here as ``torch.autograd.Function``s:

* ``_Linear``: nn.Linear over the weight's column blocks (a Linear over a ``torch.cat`` is the sum of Linears over the pieces, so
  nothing is concatenated), optionally of ``act(X)`` with X the pre-activation -- the activation runs in the GEMM's prologue and its
  backward in the input gradient's epilogue, so ``act(X)`` is never stored;
* ``_Gather2``: the first Linear of an edge MLP, ``P[row] + Q[col] + feat . W^T``, in one write-only pass;
* ``_ActScatter``: an activation and the segment sum of its output in one pass (and one pass back);
* ``_Head`` / ``_Head2``: a scalar head ``act(X W1^T + b1) . w2`` as one node -- output from the first GEMM's accumulators, the hidden
  gradient formed inside the backward GEMM kernels;
* ``_GatherAdd`` / ``_ScatterAdd`` / ``_RowScale`` / ``_Act``: row gathers, segment sums (runs of equal targets), gates, plain activations.

The edges are put in row order once per forward and the column sums go through a sorting permutation, so every edge-sized sum is a
sum over runs.  What stays in torch is the elementwise 3-vector geometry (``[E,3]``, ``[N,3,C]``, ``[B,C,C]`` tensors) and views.
``FASTEGNN_WIDE_FUSE=0`` runs every activation, head and segment sum as its own launch (A/B lever).  DESIGN.md section 9 prices the
path.  No CPU fallback: the library is loaded on first use and its absence raises."""
from __future__ import annotations

import os

import torch

from . import _lib as K
from .model import _stream

MAX_WIDE = 256


class HipOps:
    """The operators on libfastegnn_hip.so (the product path).  tests/test_wide_cpu.py drives the same orchestration through a
    torch restatement of these calls on CPU (test infrastructure, as tests/cpu_stage_backend.py does for the sharded path)."""

    @staticmethod
    def call(name, *args):
        dev = next(a.device for a in args if torch.is_tensor(a))
        K.check(getattr(K.lib(), "fastegnn_wide_" + name)(*[K.ptr(a) if torch.is_tensor(a) or a is None else a for a in args],
                                                          _stream(dev)), "fastegnn_wide_" + name)


_OPS = HipOps


def _call(name, *args):
    _OPS.call(name, *args)


def _f32(t):
    return t.contiguous().float()


class _Linear(torch.autograd.Function):
    """out = (base) + act(X) . W[:, c0:c0+K]^T (+ bias).  `act` = (kind, p) makes X the PRE-activation of the Linear's input: the
    activation runs in the GEMM's prologue (act(X) is never stored), its backward in the epilogue of the input gradient."""

    @staticmethod
    def forward(ctx, X, W, c0, Kc, bias, base, act):
        X, W = _f32(X), W.contiguous()
        M, O = X.size(0), W.size(0)
        out = torch.empty(M, O, dtype=torch.float32, device=X.device)
        b = _f32(bias) if bias is not None else None
        bs = _f32(base) if base is not None else None
        kind, p = act if act is not None else (K.ACT_NONE, 0.0)
        _call("linear", X, M, Kc, W, W.size(1), c0, b, bs, out, O, kind, p)
        ctx.save_for_backward(X, W)
        ctx.meta = (c0, Kc, bias is not None, base is not None, act)
        return out

    @staticmethod
    def backward(ctx, g):
        X, W = ctx.saved_tensors
        c0, Kc, has_bias, has_base, act = ctx.meta
        kind, p = act if act is not None else (K.ACT_NONE, 0.0)
        g = _f32(g)
        M, O = g.shape
        gX = gW = gb = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty(M, Kc, dtype=torch.float32, device=g.device)
            _call("linear_dx", g, M, O, W, W.size(1), c0, Kc, gX, 0, X if act is not None else None, kind, p)
        want_w, want_b = ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[4]
        if want_w or want_b:
            gW = torch.zeros_like(W) if want_w else None
            gb = torch.zeros(O, dtype=torch.float32, device=g.device) if want_b else None
            _call("linear_dw", g, X, M, O, Kc, gW, W.size(1), c0, gb, kind, p)
        return gX, gW, None, None, gb, (g if has_base else None), None


class _Head(torch.autograd.Function):
    """s = act(X . W1^T + b1) . w2^T (+ b2), a scalar head (coord_mlp_* / gravity_mlp, models/FastEGNN.py:55-99) as ONE autograd node:
    the backward forms the gradient of the hidden pre-activation, gs[m] * w2[o] * act'(Zc[m, o]), inside the two GEMM kernels
    (fastegnn_wide_head_dx / _dw) instead of writing it to memory and reading it back twice."""

    @staticmethod
    def forward(ctx, X, W1, b1, w2, b2, act):
        X, W1, w2 = _f32(X), W1.contiguous(), w2.contiguous()
        M, Kx, O = X.size(0), X.size(1), W1.size(0)
        zc = torch.empty(M, O, dtype=torch.float32, device=X.device)
        s = torch.empty(M, 1, dtype=torch.float32, device=X.device)
        _call("head_forward", X, M, Kx, W1, W1.size(1), 0, _f32(b1), w2, _f32(b2) if b2 is not None else None, zc, s, O, act[0], act[1],
              K.ACT_NONE, 0.0)
        ctx.save_for_backward(X, W1, w2, zc)
        ctx.meta = (act, b2 is not None)
        return s

    @staticmethod
    def backward(ctx, gs):
        X, W1, w2, zc = ctx.saved_tensors
        (kind, p), has_b2 = ctx.meta
        gs = _f32(gs)
        M, Kx, O = X.size(0), X.size(1), W1.size(0)
        dev = gs.device
        gw2 = torch.zeros_like(w2)
        gb2 = gs.sum().reshape(1) if has_b2 else None
        gX = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty(M, Kx, dtype=torch.float32, device=dev)
            _call("head_dx", gs, w2, zc, M, O, W1, W1.size(1), 0, Kx, gX, 0, kind, p)
        gW1 = torch.zeros_like(W1)
        gb1 = torch.zeros(O, dtype=torch.float32, device=dev)
        _call("head_dw", gs, w2, zc, X, M, O, Kx, gW1, W1.size(1), 0, gb1, gw2, kind, p, K.ACT_NONE, 0.0)   # (gw2 from the same pass)
        return gX, gW1, gb1, gw2, gb2, None


class _Head2(torch.autograd.Function):
    """two scalar heads over the SAME input (coord_mlp_r_virtual and coord_mlp_v_virtual over the virtual messages,
    models/FastEGNN.py:136-151): as _Head, with the second head's input gradient accumulated into the first's buffer by the kernel
    (autograd would write both and add them: three more passes over an [N*C, H] tensor)."""

    @staticmethod
    def forward(ctx, X, Wa, ba, wa2, Wb, bb, wb2, act):
        X = _f32(X)
        M, Kx = X.shape
        outs, saved = [], [X]
        for W1, b1, w2 in ((Wa, ba, wa2), (Wb, bb, wb2)):
            W1, w2 = W1.contiguous(), w2.contiguous()
            O = W1.size(0)
            zc = torch.empty(M, O, dtype=torch.float32, device=X.device)
            s = torch.empty(M, 1, dtype=torch.float32, device=X.device)
            _call("head_forward", X, M, Kx, W1, W1.size(1), 0, _f32(b1), w2, None, zc, s, O, act[0], act[1], K.ACT_NONE, 0.0)
            outs.append(s)
            saved += [W1, w2, zc]
        ctx.save_for_backward(*saved)
        ctx.act = act
        return tuple(outs)

    @staticmethod
    def backward(ctx, ga, gb):
        X = ctx.saved_tensors[0]
        kind, p = ctx.act
        M, Kx = X.shape
        dev = X.device
        gX = torch.empty(M, Kx, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        res = []
        for i, gs in enumerate((ga, gb)):
            W1, w2, zc = ctx.saved_tensors[1 + 3 * i:4 + 3 * i]
            gs = _f32(gs)
            O = W1.size(0)
            gw2 = torch.zeros_like(w2)
            if gX is not None:
                _call("head_dx", gs, w2, zc, M, O, W1, W1.size(1), 0, Kx, gX, i, kind, p)
            gW1 = torch.zeros_like(W1)
            gb1 = torch.zeros(O, dtype=torch.float32, device=dev)
            _call("head_dw", gs, w2, zc, X, M, O, Kx, gW1, W1.size(1), 0, gb1, gw2, kind, p, K.ACT_NONE, 0.0)
            res += [gW1, gb1, gw2]
        return (gX, *res, None)


def _deep_width(n):
    """csrc/wide.hip's rule for the GEMM kernel's four-buffer form: 32-column groups padded to a multiple of four cost at most a third more"""
    q = -(-n // 32)
    return 3 * (-(-q // 4) * 4) <= 4 * q


def _head_fits(X, W1):
    """the fused head takes hidden and input widths that are multiples of 4 on the GEMM kernel's four-buffer form (96, 128, 192 .. 256,
    multiples of 128): fastegnn_wide_head_dx exists in that form only"""
    O, Kx = W1.size(0), X.size(1)
    return FUSE_ACT and O % 4 == 0 and Kx % 4 == 0 and W1.size(1) == Kx and _deep_width(O) and _deep_width(Kx)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, kind, p):
        z = _f32(z)
        y = torch.empty_like(z)
        _call("act", z, z.numel(), kind, p, y)
        ctx.save_for_backward(z)
        ctx.meta = (kind, p)
        return y

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        g = _f32(g)
        dz = torch.empty_like(z)
        _call("act_backward", z, g, z.numel(), ctx.meta[0], ctx.meta[1], dz)
        return dz, None, None


class _GatherAdd(torch.autograd.Function):
    """out[m] = (base[m]) + X[idx[m]]"""

    @staticmethod
    def forward(ctx, X, idx, base):
        X = _f32(X)
        M, W = idx.numel(), X.size(1)
        out = torch.empty(M, W, dtype=torch.float32, device=X.device)
        bs = _f32(base) if base is not None else None
        _call("gather_add", X, idx, M, W, bs, out)
        ctx.save_for_backward(idx)
        ctx.meta = (X.size(0), base is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _f32(g)
        gX = None
        if ctx.needs_input_grad[0]:
            gX = torch.zeros(ctx.meta[0], g.size(1), dtype=torch.float32, device=g.device)
            _call("scatter_add", gX, idx, idx.numel(), g.size(1), g)
        return gX, None, (g if ctx.meta[1] else None)


def _scatter(table, idx, rows, order=None):
    """table[idx[m]] += rows[m]; `order` = (idx_sorted, perm) of an index in no particular order: its sums as sorted runs"""
    if order is not None and rows.size(1) >= 32:
        _call("scatter_add_perm", table, order[0], order[1], idx.numel(), rows.size(1), rows)
    else:
        _call("scatter_add", table, idx, idx.numel(), rows.size(1), rows)


class _Gather2(torch.autograd.Function):
    """out[m] = P[i1[m]] + Q[i2[m]] + feat[m] . W[:, c0:c0+nf]^T  -- the first Linear of edge_model / edge_mode_virtual over its
    torch.cat input (models/FastEGNN.py:102-119) once the node-sized products P, Q exist: one pass that only writes [M, H].
    Q / i2 and feat may be None; order2 = (sorted i2, its permutation) lets the backward sum Q's gradient as runs."""

    @staticmethod
    def forward(ctx, P, i1, Q, i2, feat, W, c0, order2):
        P = _f32(P)
        Q = _f32(Q) if Q is not None else None
        feat = _f32(feat) if feat is not None else None
        W = W.contiguous()
        M, Wd = i1.numel(), P.size(1)
        nf = feat.size(1) if feat is not None else 0
        out = torch.empty(M, Wd, dtype=torch.float32, device=P.device)
        _call("gather2", P, i1, Q, i2, feat, nf, W, W.size(1), c0, None, out, M, Wd)
        ctx.save_for_backward(i1, i2, feat, W, *(order2 if order2 is not None else ()))
        ctx.meta = (P.size(0), Q.size(0) if Q is not None else 0, c0, nf, order2 is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        i1, i2, feat, W = ctx.saved_tensors[:4]
        nP, nQ, c0, nf, has_order = ctx.meta
        order2 = tuple(ctx.saved_tensors[4:6]) if has_order else None
        g = _f32(g)
        M, Wd = g.shape
        gP = gQ = gfeat = gW = None
        if ctx.needs_input_grad[0]:
            gP = torch.zeros(nP, Wd, dtype=torch.float32, device=g.device)
            _scatter(gP, i1, g)
        if i2 is not None and ctx.needs_input_grad[2]:
            gQ = torch.zeros(nQ, Wd, dtype=torch.float32, device=g.device)
            _scatter(gQ, i2, g, order2)
        if feat is not None:
            if ctx.needs_input_grad[4]:
                gfeat = torch.empty(M, nf, dtype=torch.float32, device=g.device)
                _call("linear_dx", g, M, Wd, W, W.size(1), c0, nf, gfeat, 0, None, K.ACT_NONE, 0.0)
            if ctx.needs_input_grad[5]:
                gW = torch.zeros_like(W)
                _call("linear_dw", g, feat, M, Wd, nf, gW, W.size(1), c0, None, K.ACT_NONE, 0.0)
        return gP, None, gQ, None, gfeat, gW, None, None


class _ActScatter(torch.autograd.Function):
    """(y, table) = (act(z), segment sum of y's rows by idx into R rows) in one pass; the backward is one pass too:
    dz = (g_y + g_table[idx]) * act'(z)."""

    @staticmethod
    def forward(ctx, z, idx, R, act):
        z = _f32(z)
        y = torch.empty_like(z)
        table = torch.zeros(R, z.size(1), dtype=torch.float32, device=z.device)
        _call("act_scatter", z, idx, idx.numel(), z.size(1), act[0], act[1], y, table)
        ctx.save_for_backward(z, idx)
        ctx.act = act
        return y, table

    @staticmethod
    def backward(ctx, gy, gt):
        z, idx = ctx.saved_tensors   # (autograd hands zeros for an output nobody used)
        dz = torch.empty_like(z)
        _call("act_scatter_backward", z, idx, idx.numel(), z.size(1), ctx.act[0], ctx.act[1], _f32(gy), _f32(gt), dz)
        return dz, None, None, None


class _ScatterAdd(torch.autograd.Function):
    """table[idx[m]] += rows[m] into a zeroed [R, W] table"""

    @staticmethod
    def forward(ctx, rows, idx, R):
        rows = _f32(rows)
        table = torch.zeros(R, rows.size(1), dtype=torch.float32, device=rows.device)
        _call("scatter_add", table, idx, idx.numel(), rows.size(1), rows)
        ctx.save_for_backward(idx)
        return table

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _f32(g)
        out = torch.empty(idx.numel(), g.size(1), dtype=torch.float32, device=g.device)
        _call("gather_add", g, idx, idx.numel(), g.size(1), None, out)
        return out, None, None


class _RowScale(torch.autograd.Function):
    """Y[m, :] = X[m, :] * s[m]"""

    @staticmethod
    def forward(ctx, X, s):
        X, s = _f32(X), _f32(s)
        Y = torch.empty_like(X)
        _call("rowscale", X, s, X.size(0), X.size(1), Y)
        ctx.save_for_backward(X, s)
        return Y

    @staticmethod
    def backward(ctx, g):
        X, s = ctx.saved_tensors
        g = _f32(g)
        gX = gs = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty_like(X)
            _call("rowscale", g, s, X.size(0), X.size(1), gX)
        if ctx.needs_input_grad[1]:
            gs = torch.empty(X.size(0), dtype=torch.float32, device=g.device)
            _call("rowdot", g, X, X.size(0), X.size(1), gs)
        return gX, gs


def _rowscale(X, s):
    return _RowScale.apply(X, s.reshape(-1))


# FASTEGNN_WIDE_FUSE=0: every activation as its own launch (the unfused form of round 4; A/B lever)
FUSE_ACT = os.environ.get("FASTEGNN_WIDE_FUSE", "1") != "0"


def _lin(X, W, c0=0, Kc=None, bias=None, base=None, act=None):
    """act = (kind, p): the Linear of act(X), X the pre-activation"""
    Kc = W.size(1) - c0 if Kc is None else Kc
    if act is not None and not FUSE_ACT:
        return _Linear.apply(_Act.apply(X, act[0], act[1]), W, c0, Kc, bias, base, None)
    return _Linear.apply(X, W, c0, Kc, bias, base, act)


def _rows(X, idx):
    """X[idx] for a [R, w] table (also the 3-vector geometry: torch's own index backward sorts the indices per call)"""
    return _GatherAdd.apply(X, idx, None)


def _segment_sum(t, idx, R):
    return _ScatterAdd.apply(t, idx, R)


_WARNED_DET = False


def _warn_deterministic(model):
    """the wide path's segment sums and weight gradients are fp32 atomics in arrival order: say so once when the module was asked for
    reproducible sums (`deterministic = True` is honoured by the fused kernels only)"""
    global _WARNED_DET
    # FastEGNN / FastRF keep the request in `_deterministic` (None: not asked), EGNN in a plain `deterministic` attribute
    asked = model._deterministic if hasattr(model, "_deterministic") else getattr(model, "deterministic", False)
    if asked is True and not _WARNED_DET:
        import warnings
        warnings.warn("fastegnn_amd: hidden_nf > 64 (or EGNN flat=True) runs on the unfused wide path, whose sums are fp32 atomics: "
                      "results vary at rounding level from run to run although deterministic=True was requested", RuntimeWarning, stacklevel=3)
        _WARNED_DET = True


def forward(model, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None, node_attr=None):
    """FastEGNN.forward (models/FastEGNN.py:255-276) for hidden_nf > 64 -> (node_loc, virtual_node_loc)."""
    _warn_deterministic(model)
    dev = node_loc.device
    Hn, C = model.hidden_nf, model.virtual_channels
    kind, p = model._act
    N, B = node_loc.size(0), loc_mean.size(0)
    act = lambda z: _Act.apply(z, kind, p)                       # noqa: E731
    A = (kind, p)                                                # _lin(Z, ..., act=A): the Linear of act(Z), fused
    # the edges in row order (stable): nothing edge-sized leaves this function, and every sum over an edge's row is then a sum over
    # RUNS of equal targets (one atomic per run in fastegnn_wide_scatter_add instead of one per edge)
    row, eperm = torch.sort(edge_index[0].contiguous().long(), stable=True)
    col = edge_index[1].contiguous().long()[eperm]
    if edge_attr is not None:
        edge_attr = edge_attr[eperm]
    batch = data_batch.contiguous().long()
    ones = dict(dtype=torch.float32, device=dev)
    inv_cnt_row = 1.0 / torch.zeros(N, **ones).index_add_(0, row, torch.ones(row.numel(), **ones)).clamp(min=1)
    inv_cnt_b = 1.0 / torch.zeros(B, **ones).index_add_(0, batch, torch.ones(N, **ones)).clamp(min=1)
    idx_n = torch.arange(N, device=dev).repeat_interleave(C)                       # row n*C + c -> n
    col_sorted, col_perm = torch.sort(col, stable=True)          # once per graph: the column sums of every layer's backward as runs
    col_order = (col_sorted, col_perm)
    gravity = torch.tensor(model.gravity, **ones) if model.gravity is not None else None
    coords_sum = bool(getattr(model, "_extra_flags", 0) & K.F_COORDS_SUM)         # E_GCL_vel(coords_agg='sum'), :126
    rf = bool(getattr(model, "_extra_flags", 0) & K.F_RF)   # FastRF (models/FastRF.py:155-186): no node_model / node_model_virtual,
    #                                                         the velocity head reads ||vel|| (detached, :169) instead of h

    x, vel = node_loc.float(), node_vel.float()
    Z = loc_mean.float()                                                            # virtual_node_loc [B, 3, C]
    # virtual_node_feat [B, H, C] is kept as rows (b, c) of width H
    HvT = model.virtual_node_feat.repeat(B, 1, 1).permute(0, 2, 1).reshape(B * C, Hn)
    h = _lin(node_feat.float(), model.embedding_in.weight, 0, model.node_feat_nf, model.embedding_in.bias)

    def head(seq, X):   # coord_mlp: Linear(H, H), act, Linear(H, 1, bias=False) [, Tanh]   (:55-67)
        if _head_fits(X, seq[0].weight):
            s = _Head.apply(X, seq[0].weight, seq[0].bias, seq[2].weight, None, A)
        else:
            s = _lin(_lin(X, seq[0].weight, 0, Hn, seq[0].bias), seq[2].weight, 0, Hn, None, None, A)
        return torch.tanh(s) if model.tanh else s

    def scalar_head_in(seq, X):   # Linear(w, H), act, Linear(H, 1) over an input of any width w
        if _head_fits(X, seq[0].weight):
            return _Head.apply(X, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, A)
        return _lin(_lin(X, seq[0].weight, 0, X.size(1), seq[0].bias), seq[2].weight, 0, Hn, seq[2].bias, None, A)

    def scalar_head(seq, X):   # Linear(H, H), act, Linear(H, 1)   (:75-88)
        return scalar_head_in(seq, X)

    for i in range(model.n_layers):
        g = getattr(model, "gcl_%d" % i)
        # ---- coord2radial (:176-185) and the virtual geometry (:200-201): 3-vectors, torch
        cd = _rows(x, row) - _rows(x, col)
        radial = (cd * cd).sum(1, keepdim=True)
        if model.normalize:
            cd = cd / (torch.sqrt(radial).detach() + 1e-8)
        vcd = _rows(Z.reshape(B, 3 * C), batch).view(N, 3, C) - x.unsqueeze(-1)      # [N, 3, C]
        vr = torch.norm(vcd, p=2, dim=1, keepdim=True)                              # [N, 1, C]
        # ---- edge_model (:102-108): Linear over cat[h[row], h[col], radial, edge_attr] = P[row] + Q[col] + feat . W[:, 2H:]
        W1 = g.edge_mlp[0].weight
        feat = radial if edge_attr is None else torch.cat([radial, edge_attr.float()], 1)
        if feat.size(1) <= 8:
            pre = _Gather2.apply(_lin(h, W1, 0, Hn, g.edge_mlp[0].bias), row, _lin(h, W1, Hn, Hn), col, feat, W1, 2 * Hn, col_order)
        else:   # (more than 7 edge attributes: the feature columns as a product of their own)
            pre = _Gather2.apply(_lin(h, W1, 0, Hn, g.edge_mlp[0].bias), row, _lin(h, W1, Hn, Hn), col, None, W1, 0, col_order)
            pre = _lin(feat, W1, 2 * Hn, feat.size(1), None, pre)
        z2 = _lin(pre, g.edge_mlp[2].weight, 0, Hn, g.edge_mlp[2].bias, None, A)
        fuse_sum = FUSE_ACT and not model.attention and not rf   # act and node_model's segment sum of it in one pass
        if fuse_sum:
            m, agg_m = _ActScatter.apply(z2, row, N, A)                              # [E, H], [N, H]
        else:
            m = act(z2)                                                              # [E, H]
        if model.attention:
            m = _rowscale(m, torch.sigmoid(_lin(m, g.att_mlp[0].weight, 0, Hn, g.att_mlp[0].bias)))
        # ---- edge_mode_virtual (:111-119): rows (n, c); input cat[h, Hv[b], vr, m_X[b][:, c]]
        cm = _segment_sum(x, batch, B) * inv_cnt_b.unsqueeze(1)               # global_mean_pool(coord)
        mX = Z - cm.unsqueeze(-1)
        mX = torch.einsum('bij,bjk->bik', mX.permute(0, 2, 1), mX)                  # [B, C, C]
        Wv = g.edge_mlp_virtual[0].weight
        Bc = _lin(mX.permute(0, 2, 1).reshape(B * C, C), Wv, 2 * Hn + 1, C, None, _lin(HvT, Wv, Hn, Hn))
        # rows (n, c) <- A[n] + Bc[b(n), c]: the second gather moves whole [C*H] rows by graph (its adjoint is then a segment sum
        # over sorted indices instead of N*C atomics onto B*C rows)
        pv = _Gather2.apply(_lin(h, Wv, 0, Hn, g.edge_mlp_virtual[0].bias), idx_n, None, None, vr.reshape(N * C, 1), Wv, 2 * Hn, None)
        pv = _GatherAdd.apply(Bc.view(B, C * Hn), batch, pv.view(N, C * Hn)).view(N * C, Hn)
        z2v = _lin(pv, g.edge_mlp_virtual[2].weight, 0, Hn, g.edge_mlp_virtual[2].bias, None, A)
        if fuse_sum:   # rows (n, c) of a node are one [C*H] row of graph batch[n]: node_model_virtual's pool as sorted runs
            v, pool_v = _ActScatter.apply(z2v.view(N, C * Hn), batch, B, A)
            v = v.view(N * C, Hn)
        else:
            v = act(z2v)                                                             # [N*C, H]
        if model.attention:
            v = _rowscale(v, torch.sigmoid(_lin(v, g.att_mlp_virtual[0].weight, 0, Hn, g.att_mlp_virtual[0].bias)))
        # ---- coord_model_vel (:122-145)
        trans = cd * head(g.coord_mlp_r, m)
        agg = _segment_sum(trans, row, N)
        x_new = x + (agg if coords_sum else agg * inv_cnt_row.unsqueeze(1))
        hr, hv = g.coord_mlp_r_virtual, g.coord_mlp_v_virtual
        if _head_fits(v, hr[0].weight) and _head_fits(v, hv[0].weight):   # both heads of the virtual messages as one node
            s_rv, s_vv = _Head2.apply(v, hr[0].weight, hr[0].bias, hr[2].weight, hv[0].weight, hv[0].bias, hv[2].weight, A)
            if model.tanh:
                s_rv, s_vv = torch.tanh(s_rv), torch.tanh(s_vv)
        else:
            s_rv, s_vv = head(hr, v), head(hv, v)
        x_new = x_new + torch.mean(-vcd * s_rv.reshape(N, 1, C), dim=-1)
        if rf:
            x_new = x_new + scalar_head_in(g.coord_mlp_vel, torch.norm(vel, p=2, dim=-1).unsqueeze(-1).detach()) * vel
        else:
            x_new = x_new + scalar_head(g.coord_mlp_vel, h) * vel
        if gravity is not None:
            x_new = x_new + scalar_head(g.gravity_mlp, h) * gravity
        # ---- coord_model_virtual (:147-151)
        transX = vcd * s_vv.reshape(N, 1, C)
        Z_new = Z + (_segment_sum(transX.reshape(N, 3 * C), batch, B) * inv_cnt_b.unsqueeze(1)).reshape(B, 3, C)
        if rf:   # the features of real and virtual nodes pass through (FastRF.py:186)
            x, Z = x_new, Z_new
            continue
        # ---- node_model (:154-166): Linear over cat[h, agg, flat(v), node_attr]; flat(v) of the reference is (h, c)-ordered
        aggm = _rowscale(agg_m if fuse_sum else _ScatterAdd.apply(m, row, N), inv_cnt_row)
        W3 = g.node_mlp[0].weight
        W3v = W3[:, 2 * Hn:2 * Hn + Hn * C].reshape(W3.size(0), Hn, C).permute(0, 2, 1).reshape(W3.size(0), C * Hn)
        npre = _lin(aggm, W3, Hn, Hn, None, _lin(h, W3, 0, Hn, g.node_mlp[0].bias))
        npre = _lin(v.view(N, C * Hn), W3v, 0, C * Hn, None, npre)
        if node_attr is not None:
            npre = _lin(node_attr.float(), W3, 2 * Hn + Hn * C, node_attr.size(1), None, npre)
        h_new = _lin(npre, g.node_mlp[2].weight, 0, Hn, g.node_mlp[2].bias, h if model.residual else None, A)
        # ---- node_model_virtual (:168-178)
        poolV = _rowscale(pool_v if fuse_sum else _ScatterAdd.apply(v.view(N, C * Hn), batch, B), inv_cnt_b).view(B * C, Hn)
        Wn = g.node_mlp_virtual[0].weight
        zv = _lin(poolV, Wn, Hn, Hn, None, _lin(HvT, Wn, 0, Hn, g.node_mlp_virtual[0].bias))
        HvT = _lin(zv, g.node_mlp_virtual[2].weight, 0, Hn, g.node_mlp_virtual[2].bias, HvT if model.residual else None, A)
        h, x, Z = h_new, x_new, Z_new
    return x, Z


def egnn_forward(model, x, h, edge_index, edge_fea, v=None):
    """EGNN.forward (models/basic.py:337-341 over EGNN_Layer.forward :302-320) on the wide operators: hidden_nf > 64, or
    flat=True (every BaseMLP a Tanh MLP with 4 x hidden inner units, :176-178) -> (x, h)."""
    _warn_deterministic(model)
    dev = x.device
    Hn = model.hidden_nf
    kind, p = (K.ACT_TANH, 0.0) if model.flat else model._act
    act = lambda z: _Act.apply(z, kind, p)                       # noqa: E731
    A = (kind, p)
    N = x.size(0)
    row, eperm = torch.sort(edge_index[0].contiguous().long(), stable=True)   # edges in row order: sums over runs (see forward)
    col = edge_index[1].contiguous().long()[eperm]
    if edge_fea is not None:
        edge_fea = edge_fea[eperm]
    f32 = dict(dtype=torch.float32, device=dev)
    inv_cnt = 1.0 / torch.zeros(N, **f32).index_add_(0, row, torch.ones(row.numel(), **f32)).clamp(min=1)   # aggregate(aggr='mean'), :27-52
    col_order = tuple(torch.sort(col, stable=True))
    x = x.float()
    vv = v.float() if v is not None else None
    h = _lin(h.float(), model.embedding.weight, 0, model.in_node_nf, model.embedding.bias)

    def mlp(net, X, base_first=None):   # BaseMLP without last_act: Linear, act, Linear
        if base_first is None and net.mlp[2].weight.size(0) == 1 and _head_fits(X, net.mlp[0].weight):   # coord_net / node_v_net: scalar heads
            return _Head.apply(X, net.mlp[0].weight, net.mlp[0].bias, net.mlp[2].weight, net.mlp[2].bias, A)
        return _lin(_lin(X, net.mlp[0].weight, 0, X.size(1), net.mlp[0].bias) if base_first is None else base_first,
                    net.mlp[2].weight, 0, net.mlp[2].weight.size(1), net.mlp[2].bias, None, A)

    for layer in model.layers:
        rij = _rows(x, row) - _rows(x, col)
        scalar = (rij * rij).sum(1, keepdim=True)                                   # 1 x 1 Gram of the single vector, :268-270
        if model.norm:
            scalar = torch.nn.functional.normalize(scalar, p=2, dim=-1)
        # edge_message_net: BaseMLP(last_act=True) over cat(scalar, h[row], h[col], edge_fea)  (:313, :259-261)
        net = layer.edge_message_net.scalar_net.mlp
        W0 = net[0].weight
        pre = _Gather2.apply(_lin(h, W0, 1, Hn, net[0].bias), row, _lin(h, W0, 1 + Hn, Hn), col, scalar, W0, 0, col_order)
        if edge_fea is not None:
            pre = _lin(edge_fea.float(), W0, 1 + 2 * Hn, edge_fea.size(1), None, pre)
        zm = _lin(pre, net[2].weight, 0, net[2].weight.size(1), net[2].bias, None, A)
        if FUSE_ACT:
            message, sum_message = _ActScatter.apply(zm, row, N, A)                   # [E, H], [N, H]
        else:
            message = act(zm)
        f = rij * mlp(layer.coord_net, message)
        tot_f = torch.clamp(_segment_sum(f, row, N) * inv_cnt.unsqueeze(1), min=-100, max=100)
        x_new = x + tot_f
        if vv is not None:
            x_new = x_new + mlp(layer.node_v_net, h) * vv
        tot_message = _rowscale(sum_message if FUSE_ACT else _ScatterAdd.apply(message, row, N), inv_cnt)
        Wn = layer.node_net.mlp[0].weight
        h = mlp(layer.node_net, None, _lin(tot_message, Wn, Hn, Hn, None, _lin(h, Wn, 0, Hn, layer.node_net.mlp[0].bias)))
        x = x_new
    return x, h
