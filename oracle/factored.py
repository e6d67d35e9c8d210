"""TEST INFRASTRUCTURE ONLY -- stage-level CPU spec of the HIP kernels.

The product (``fastegnn_amd/csrc``) evaluates one E_GCL_vel layer
(``/root/reference/models/FastEGNN.py:192-223``) in a *factorised* form: the
first Linear of ``edge_mlp`` / ``edge_mlp_virtual`` is split by input block
(SURVEY.md section 3.5), activations are never concatenated, and the backward
is hand-derived.  This file states each stage (same names and buffers as
``include/fastegnn_hip.h``) with plain torch ops so that

  * ``tests/test_factored_cpu.py`` can check the hand-derived backward against
    autograd of the op-for-op oracle (``oracle/fastegnn_ref.py``) in fp64, and
  * the GPU tests can diff every HIP stage against its CPU statement, and
  * the world_size-2 gloo test can drive the sharded orchestration on CPU.

It is dtype-generic (fp32 or fp64).  Not imported by the product.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .fastegnn_ref import Config, Params

H = 64

# bf16 operand mode (Config.bf16; the product's FASTEGNN_F_BF16): BOTH operands of every 64-wide contraction that the
# kernels run on the matrix cores -- the [64,64] weights and the 64-column blocks of edge_mlp.0 / edge_mlp_virtual.0 /
# node_mlp.0 / node_mlp_virtual.0, in the forward products, the transposed (input-gradient) products and the weight-
# gradient products -- are rounded to bf16 (round to nearest even) and multiplied exactly with fp32 (here: the working
# dtype's) accumulation.  Everything else stays in the working precision: coordinates, radial / vr / Gram terms and
# their weight columns, edge_attr / node_attr columns, the [1,64] heads, attention gates, biases, activations,
# segment sums and pools (hard part 7).
_BF16 = False


def R(t):
    """operand rounding of the bf16 mode (identity otherwise)."""
    return t.to(torch.bfloat16).to(t.dtype) if _BF16 else t


class _mode:
    def __init__(self, cfg):
        self.on = bool(getattr(cfg, "bf16", False))

    def __enter__(self):
        global _BF16
        self.prev, _BF16 = _BF16, self.on

    def __exit__(self, *a):
        global _BF16
        _BF16 = self.prev


def silu(z):
    return z * torch.sigmoid(z)


def dsilu(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


# --------------------------------------------------------------------------
# graph bookkeeping (the C-ABI's fastegnn_build_csr)
# --------------------------------------------------------------------------
@dataclass
class Csr:
    n: int
    rowptr: torch.Tensor      # int64 [N+1]
    row: torch.Tensor         # int64 [E]  row of each sorted edge
    col: torch.Tensor         # int64 [E]
    perm: torch.Tensor        # int64 [E]  sorted edge k = input edge perm[k]
    cscptr: torch.Tensor      # int64 [N+1]
    csc_eid: torch.Tensor     # int64 [E]  sorted-edge ids grouped by col
    inv_deg: torch.Tensor     # [N] 1/max(deg,1)


def build_csr(edge_index, n, dtype=torch.float32) -> Csr:
    row, col = edge_index[0], edge_index[1]
    perm = torch.sort(row, stable=True).indices
    r, c = row[perm], col[perm]
    deg = torch.bincount(r, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(deg, 0)
    csc_eid = torch.sort(c, stable=True).indices
    cdeg = torch.bincount(c, minlength=n)
    cscptr = torch.zeros(n + 1, dtype=torch.long)
    cscptr[1:] = torch.cumsum(cdeg, 0)
    return Csr(n, rowptr, r, c, perm, cscptr, csc_eid, 1.0 / deg.clamp(min=1).to(dtype))


def graph_ptr(batch, B):
    cnt = torch.bincount(batch, minlength=B)
    ptr = torch.zeros(B + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(cnt, 0)
    return ptr


# --------------------------------------------------------------------------
# per-layer weight views (slices of the reference state_dict tensors)
# --------------------------------------------------------------------------
class LayerW:
    def __init__(self, p: Params, L: str, cfg: Config):
        C, ea, na = cfg.virtual_channels, cfg.edge_attr_nf, cfg.node_attr_nf
        g = lambda k: p[f"{L}.{k}"]
        W1 = g("edge_mlp.0.weight")
        self.W1a, self.W1b, self.w_r, self.We = W1[:, :H], W1[:, H:2 * H], W1[:, 2 * H], W1[:, 2 * H + 1:]
        self.b1 = g("edge_mlp.0.bias")
        self.W2, self.b2 = g("edge_mlp.2.weight"), g("edge_mlp.2.bias")
        V1 = g("edge_mlp_virtual.0.weight")
        self.V1a, self.V1b, self.w_vr, self.V1d = V1[:, :H], V1[:, H:2 * H], V1[:, 2 * H], V1[:, 2 * H + 1:]
        self.c1 = g("edge_mlp_virtual.0.bias")
        self.V2, self.c2 = g("edge_mlp_virtual.2.weight"), g("edge_mlp_virtual.2.bias")
        if cfg.attention:
            self.att_w, self.att_b = g("att_mlp.0.weight")[0], g("att_mlp.0.bias")[0]
            self.attv_w, self.attv_b = g("att_mlp_virtual.0.weight")[0], g("att_mlp_virtual.0.bias")[0]
        self.Wx1, self.bx1, self.wx2 = g("coord_mlp_r.0.weight"), g("coord_mlp_r.0.bias"), g("coord_mlp_r.2.weight")[0]
        self.Wxv0, self.bxv0, self.wxv2 = (g("coord_mlp_r_virtual.0.weight"), g("coord_mlp_r_virtual.0.bias"),
                                           g("coord_mlp_r_virtual.2.weight")[0])
        self.WX0, self.bX0, self.wX2 = (g("coord_mlp_v_virtual.0.weight"), g("coord_mlp_v_virtual.0.bias"),
                                        g("coord_mlp_v_virtual.2.weight")[0])
        self.Wv0, self.bv0 = g("coord_mlp_vel.0.weight"), g("coord_mlp_vel.0.bias")
        self.wv2, self.bv2 = g("coord_mlp_vel.2.weight")[0], g("coord_mlp_vel.2.bias")[0]
        if cfg.gravity is not None:
            self.Wg0, self.bg0 = g("gravity_mlp.0.weight"), g("gravity_mlp.0.bias")
            self.wg2, self.bg2 = g("gravity_mlp.2.weight")[0], g("gravity_mlp.2.bias")[0]
        W3 = g("node_mlp.0.weight")
        self.W3a, self.W3b = W3[:, :H], W3[:, H:2 * H]
        # flat(v)[h*C + c]  ->  W3v[o, h, c]
        self.W3v = W3[:, 2 * H:2 * H + H * C].reshape(H, H, C)
        self.W3d = W3[:, 2 * H + H * C:] if na > 0 else None
        self.b3 = g("node_mlp.0.bias")
        self.W4, self.b4 = g("node_mlp.2.weight"), g("node_mlp.2.bias")
        W5 = g("node_mlp_virtual.0.weight")
        self.W5a, self.W5b, self.b5 = W5[:, :H], W5[:, H:], g("node_mlp_virtual.0.bias")
        self.W6, self.b6 = g("node_mlp_virtual.2.weight"), g("node_mlp_virtual.2.bias")


def _zeros_like_params(p: Params, L: str):
    return {k: torch.zeros_like(v) for k, v in p.items() if k.startswith(L + ".")}


# --------------------------------------------------------------------------
# S1 node_pre
# --------------------------------------------------------------------------
def node_pre_fwd(w: LayerW, cfg: Config, h):
    hr = R(h)
    P = hr @ R(w.W1a).T + w.b1
    Q = hr @ R(w.W1b).T
    A = hr @ R(w.V1a).T
    svel = silu(hr @ R(w.Wv0).T + w.bv0) @ w.wv2 + w.bv2
    sgrav = None
    if cfg.gravity is not None:
        sgrav = silu(hr @ R(w.Wg0).T + w.bg0) @ w.wg2 + w.bg2
    return P, Q, A, svel, sgrav


def node_pre_bwd(w: LayerW, cfg: Config, G: Dict[str, torch.Tensor], L: str, h, g_P, g_Q, g_A, g_svel, g_sgrav):
    """returns d/dh contribution; accumulates weight grads into G."""
    hr = R(h)
    g_h = R(g_P) @ R(w.W1a) + R(g_Q) @ R(w.W1b) + R(g_A) @ R(w.V1a)
    dW1 = G[f"{L}.edge_mlp.0.weight"]
    dW1[:, :H] += R(g_P).T @ hr
    dW1[:, H:2 * H] += R(g_Q).T @ hr
    G[f"{L}.edge_mlp.0.bias"] += g_P.sum(0)
    G[f"{L}.edge_mlp_virtual.0.weight"][:, :H] += R(g_A).T @ hr

    def head(W0, b0, w2, g_s, name):
        z = hr @ R(W0).T + b0
        u = silu(z)
        G[f"{L}.{name}.2.weight"][0] += g_s @ u
        G[f"{L}.{name}.2.bias"][0] += g_s.sum()
        g_z = (g_s.unsqueeze(1) * w2) * dsilu(z)
        G[f"{L}.{name}.0.weight"] += R(g_z).T @ hr
        G[f"{L}.{name}.0.bias"] += g_z.sum(0)
        return R(g_z) @ R(W0)

    g_h = g_h + head(w.Wv0, w.bv0, w.wv2, g_svel, "coord_mlp_vel")
    if cfg.gravity is not None:
        g_h = g_h + head(w.Wg0, w.bg0, w.wg2, g_sgrav, "gravity_mlp")
    return g_h


# --------------------------------------------------------------------------
# S2 graph_pre  (centroid, Gram, per-(graph,channel) first-layer term Bc)
# --------------------------------------------------------------------------
def graph_pre_fwd(w: LayerW, cfg: Config, x, gptr, Z, HvT):
    """HvT is [B,C,H] (the kernels' channel-major layout of the reference's [B,H,C])."""
    B = Z.size(0)
    n_b = (gptr[1:] - gptr[:-1]).clamp(min=1).to(x.dtype)
    xsum = torch.zeros(B, 3, dtype=x.dtype)
    batch = torch.repeat_interleave(torch.arange(B), gptr[1:] - gptr[:-1])
    xsum.index_add_(0, batch, x)
    xbar = xsum / n_b.unsqueeze(1)
    mz = Z - xbar.unsqueeze(-1)                         # [B,3,C]
    mX = torch.einsum("bkc,bkd->bcd", mz, mz)           # [B,C,C]
    # feature vector of channel c is column c of mX: mX[b][:, c]
    Bc = R(HvT) @ R(w.V1b).T + mX.transpose(1, 2) @ w.V1d.T + w.c1       # [B,C,H]
    return xbar, mX, Bc


def graph_pre_bwd(w: LayerW, cfg: Config, G, L: str, x, gptr, Z, HvT, g_Bc):
    """returns g_HvT, g_Z, g_xbar_per_graph (already divided by n_b -> add to every node)."""
    xbar, mX, _ = graph_pre_fwd(w, cfg, x, gptr, Z, HvT)
    n_b = (gptr[1:] - gptr[:-1]).clamp(min=1).to(x.dtype)
    mz = Z - xbar.unsqueeze(-1)
    C = Z.size(2)
    g2 = g_Bc.reshape(-1, H)
    dV1 = G[f"{L}.edge_mlp_virtual.0.weight"]
    dV1[:, H:2 * H] += R(g2).T @ R(HvT).reshape(-1, H)
    dV1[:, 2 * H + 1:] += g2.T @ mX.transpose(1, 2).reshape(-1, C)
    G[f"{L}.edge_mlp_virtual.0.bias"] += g2.sum(0)
    g_HvT = R(g_Bc) @ R(w.V1b)
    g_mXT = g_Bc @ w.V1d                                  # [B,C(c),C(c')] = d/d mX[b][c',c]
    g_mX = g_mXT.transpose(1, 2)
    g_mz = torch.einsum("bkd,bcd->bkc", mz, g_mX + g_mX.transpose(1, 2))
    g_Z = g_mz
    g_xbar = -g_mz.sum(-1) / n_b.unsqueeze(1)             # [B,3], per-node share
    return g_HvT, g_Z, g_xbar


# --------------------------------------------------------------------------
# S3 edge  (gather -> edge MLP -> coordinate head -> row-segmented mean)
# --------------------------------------------------------------------------
def _edge_recompute(w: LayerW, cfg: Config, csr: Csr, P, Q, x, ea, x_src=None):
    """x_src/Q may be a larger *source table* addressed by csr.col (sharded graphs); x/P are the
    tables of the aggregation rows."""
    d = x[csr.row] - (x if x_src is None else x_src)[csr.col]
    r = (d * d).sum(1)
    nrm = r.sqrt()
    dn = d / (nrm + cfg.epsilon).unsqueeze(1) if cfg.normalize else d
    pre = P[csr.row] + Q[csr.col] + r.unsqueeze(1) * w.w_r + ea @ w.We.T
    t = silu(pre)
    mp = R(t) @ R(w.W2).T + w.b2
    m0 = silu(mp)
    if cfg.attention:
        a = torch.sigmoid(m0 @ w.att_w + w.att_b)
        m = m0 * a.unsqueeze(1)
    else:
        a, m = None, m0
    up = R(m) @ R(w.Wx1).T + w.bx1
    u = silu(up)
    sraw = u @ w.wx2
    s = torch.tanh(sraw) if cfg.tanh else sraw
    return dict(d=d, r=r, nrm=nrm, dn=dn, pre=pre, t=t, mp=mp, m0=m0, a=a, m=m, up=up, u=u, s=s)


def edge_fwd(w: LayerW, cfg: Config, csr: Csr, P, Q, x, ea, x_src=None):
    """ea is in sorted-edge order.  returns aggm [N,H] (mean), aggx [N,3] (mean|sum)."""
    k = _edge_recompute(w, cfg, csr, P, Q, x, ea, x_src)
    N = csr.n
    aggm = torch.zeros(N, H, dtype=x.dtype).index_add_(0, csr.row, k["m"]) * csr.inv_deg.unsqueeze(1)
    aggx = torch.zeros(N, 3, dtype=x.dtype).index_add_(0, csr.row, k["dn"] * k["s"].unsqueeze(1))
    if cfg.coords_agg == "mean":
        aggx = aggx * csr.inv_deg.unsqueeze(1)
    return aggm, aggx


def edge_bwd(w: LayerW, cfg: Config, G, L: str, csr: Csr, P, Q, x, ea, g_aggm, g_aggx, x_src=None):
    """returns g_P [N,H], g_Q [N,H], g_x [N,3] (row side + col side); with a source table (x_src given)
    returns g_P, g_Q [n_src,H], (g_x_row [N,3], g_x_src [n_src,3])."""
    k = _edge_recompute(w, cfg, csr, P, Q, x, ea, x_src)
    N = csr.n
    idg = csr.inv_deg[csr.row]
    g_m = g_aggm[csr.row] * idg.unsqueeze(1)
    g_trans = g_aggx[csr.row] * (idg.unsqueeze(1) if cfg.coords_agg == "mean" else 1.0)
    g_s = (k["dn"] * g_trans).sum(1)
    g_dn = k["s"].unsqueeze(1) * g_trans
    g_sraw = g_s * (1 - k["s"] ** 2) if cfg.tanh else g_s
    G[f"{L}.coord_mlp_r.2.weight"][0] += g_sraw @ k["u"]
    g_up = (g_sraw.unsqueeze(1) * w.wx2) * dsilu(k["up"])
    G[f"{L}.coord_mlp_r.0.bias"] += g_up.sum(0)
    G[f"{L}.coord_mlp_r.0.weight"] += R(g_up).T @ R(k["m"])
    g_m = g_m + R(g_up) @ R(w.Wx1)
    if cfg.attention:
        a = k["a"]
        g_a = (g_m * k["m0"]).sum(1)
        g_z = g_a * a * (1 - a)
        G[f"{L}.att_mlp.0.weight"][0] += g_z @ k["m0"]
        G[f"{L}.att_mlp.0.bias"][0] += g_z.sum()
        g_m0 = g_m * a.unsqueeze(1) + g_z.unsqueeze(1) * w.att_w
    else:
        g_m0 = g_m
    g_mp = g_m0 * dsilu(k["mp"])
    G[f"{L}.edge_mlp.2.bias"] += g_mp.sum(0)
    G[f"{L}.edge_mlp.2.weight"] += R(g_mp).T @ R(k["t"])
    g_pre = (R(g_mp) @ R(w.W2)) * dsilu(k["pre"])
    dW1 = G[f"{L}.edge_mlp.0.weight"]
    dW1[:, 2 * H] += g_pre.T @ k["r"]
    dW1[:, 2 * H + 1:] += g_pre.T @ ea
    g_r = g_pre @ w.w_r
    g_d = (g_dn / (k["nrm"] + cfg.epsilon).unsqueeze(1) if cfg.normalize else g_dn) + 2 * g_r.unsqueeze(1) * k["d"]
    g_P = torch.zeros(N, H, dtype=x.dtype).index_add_(0, csr.row, g_pre)
    if x_src is not None:
        ns = x_src.size(0)
        g_Q = torch.zeros(ns, H, dtype=x.dtype).index_add_(0, csr.col, g_pre)
        g_xr = torch.zeros(N, 3, dtype=x.dtype).index_add_(0, csr.row, g_d)
        g_xs = torch.zeros(ns, 3, dtype=x.dtype).index_add_(0, csr.col, -g_d)
        return g_P, g_Q, (g_xr, g_xs)
    g_Q = torch.zeros(N, H, dtype=x.dtype).index_add_(0, csr.col, g_pre)
    g_x = torch.zeros(N, 3, dtype=x.dtype).index_add_(0, csr.row, g_d).index_add_(0, csr.col, -g_d)
    return g_P, g_Q, g_x


# --------------------------------------------------------------------------
# S4 virt  (node x channel block + node MLP + coordinate update + pools)
# --------------------------------------------------------------------------
def _virt_recompute(w: LayerW, cfg: Config, A, Bc, x, Z, batch):
    vd = Z[batch] - x.unsqueeze(-1)                       # [N,3,C]
    vr = (vd * vd).sum(1).sqrt()                          # [N,C]
    pre = A.unsqueeze(1) + Bc[batch] + vr.unsqueeze(-1) * w.w_vr      # [N,C,H]
    t = silu(pre)
    vp = R(t) @ R(w.V2).T + w.c2
    v0 = silu(vp)
    if cfg.attention:
        a = torch.sigmoid(v0 @ w.attv_w + w.attv_b)      # [N,C]
        v = v0 * a.unsqueeze(-1)
    else:
        a, v = None, v0
    uxp = R(v) @ R(w.Wxv0).T + w.bxv0
    ux = silu(uxp)
    sxr = ux @ w.wxv2
    sx = torch.tanh(sxr) if cfg.tanh else sxr
    uXp = R(v) @ R(w.WX0).T + w.bX0
    uX = silu(uXp)
    sXr = uX @ w.wX2
    sX = torch.tanh(sXr) if cfg.tanh else sXr
    return dict(vd=vd, vr=vr, pre=pre, t=t, vp=vp, v0=v0, a=a, v=v, uxp=uxp, ux=ux, sx=sx, uXp=uXp, uX=uX, sX=sX)


def virt_fwd(w: LayerW, cfg: Config, h, A, Bc, x, vel, Z, batch, aggm, aggx, svel, sgrav, gravity, node_attr=None):
    """returns h_new, x_new, poolV [B,C,H] (sum over nodes), poolX [B,3,C] (sum over nodes)."""
    B, C = Z.size(0), Z.size(2)
    k = _virt_recompute(w, cfg, A, Bc, x, Z, batch)
    transv = (-k["vd"] * k["sx"].unsqueeze(1)).mean(-1)                     # [N,3]
    poolX = torch.zeros(B, 3, C, dtype=x.dtype).index_add_(0, batch, k["vd"] * k["sX"].unsqueeze(1))
    poolV = torch.zeros(B, C, H, dtype=x.dtype).index_add_(0, batch, k["v"])
    nodepre = torch.einsum("ohc,nch->no", R(w.W3v), R(k["v"])) + R(h) @ R(w.W3a).T + R(aggm) @ R(w.W3b).T + w.b3
    if node_attr is not None:
        nodepre = nodepre + node_attr @ w.W3d.T
    out = R(silu(nodepre)) @ R(w.W4).T + w.b4
    h_new = h + out if cfg.residual else out
    x_new = x + aggx + transv + svel.unsqueeze(1) * vel
    if gravity is not None:
        x_new = x_new + sgrav.unsqueeze(1) * gravity
    return h_new, x_new, poolV, poolX


def virt_bwd(w: LayerW, cfg: Config, G, L: str, h, A, Bc, x, vel, Z, batch, aggm, gravity,
             g_hn, g_xn, g_poolV, g_poolX, node_attr=None):
    """returns dict(g_h, g_x, g_A, g_aggm, g_aggx, g_svel, g_sgrav, g_Bc, g_Z, g_vel[, via svel])."""
    B, C = Z.size(0), Z.size(2)
    N = h.size(0)
    k = _virt_recompute(w, cfg, A, Bc, x, Z, batch)
    v = k["v"]
    nodepre = torch.einsum("ohc,nch->no", R(w.W3v), R(v)) + R(h) @ R(w.W3a).T + R(aggm) @ R(w.W3b).T + w.b3
    if node_attr is not None:
        nodepre = nodepre + node_attr @ w.W3d.T
    t3 = silu(nodepre)
    # node MLP
    g_out = g_hn
    g_h = g_hn.clone() if cfg.residual else torch.zeros_like(g_hn)
    G[f"{L}.node_mlp.2.bias"] += g_out.sum(0)
    G[f"{L}.node_mlp.2.weight"] += R(g_out).T @ R(t3)
    g_np = (R(g_out) @ R(w.W4)) * dsilu(nodepre)
    dW3 = G[f"{L}.node_mlp.0.weight"]
    G[f"{L}.node_mlp.0.bias"] += g_np.sum(0)
    dW3[:, :H] += R(g_np).T @ R(h)
    dW3[:, H:2 * H] += R(g_np).T @ R(aggm)
    dW3[:, 2 * H:2 * H + H * C] += torch.einsum("no,nch->ohc", R(g_np), R(v)).reshape(H, H * C)
    if node_attr is not None:
        dW3[:, 2 * H + H * C:] += g_np.T @ node_attr
    g_h = g_h + R(g_np) @ R(w.W3a)
    g_aggm = R(g_np) @ R(w.W3b)
    g_v = torch.einsum("no,ohc->nch", R(g_np), R(w.W3v)) + g_poolV[batch]
    # coordinate update
    g_x = g_xn.clone()
    g_aggx = g_xn
    g_svel = (g_xn * vel).sum(1)
    g_sgrav = (g_xn * gravity).sum(1) if gravity is not None else None
    vd = k["vd"]
    g_sx = (-vd * g_xn.unsqueeze(-1)).sum(1) / C                           # [N,C]
    g_vd = -k["sx"].unsqueeze(1) * g_xn.unsqueeze(-1) / C                   # [N,3,C]
    gpX = g_poolX[batch]                                                   # [N,3,C]
    g_sX = (vd * gpX).sum(1)
    g_vd = g_vd + k["sX"].unsqueeze(1) * gpX

    def head(g_s, s, u, up, W0, w2, name):
        g_sr = g_s * (1 - s * s) if cfg.tanh else g_s
        G[f"{L}.{name}.2.weight"][0] += torch.einsum("nc,nch->h", g_sr, u)
        g_up = (g_sr.unsqueeze(-1) * w2) * dsilu(up)
        G[f"{L}.{name}.0.bias"] += g_up.sum((0, 1))
        G[f"{L}.{name}.0.weight"] += torch.einsum("nco,nch->oh", R(g_up), R(v))
        return R(g_up) @ R(W0)

    g_v = g_v + head(g_sx, k["sx"], k["ux"], k["uxp"], w.Wxv0, w.wxv2, "coord_mlp_r_virtual")
    g_v = g_v + head(g_sX, k["sX"], k["uX"], k["uXp"], w.WX0, w.wX2, "coord_mlp_v_virtual")
    if cfg.attention:
        a = k["a"]
        g_a = (g_v * k["v0"]).sum(-1)
        g_z = g_a * a * (1 - a)
        G[f"{L}.att_mlp_virtual.0.weight"][0] += torch.einsum("nc,nch->h", g_z, k["v0"])
        G[f"{L}.att_mlp_virtual.0.bias"][0] += g_z.sum()
        g_v0 = g_v * a.unsqueeze(-1) + g_z.unsqueeze(-1) * w.attv_w
    else:
        g_v0 = g_v
    g_vp = g_v0 * dsilu(k["vp"])
    G[f"{L}.edge_mlp_virtual.2.bias"] += g_vp.sum((0, 1))
    G[f"{L}.edge_mlp_virtual.2.weight"] += torch.einsum("nco,nch->oh", R(g_vp), R(k["t"]))
    g_pre = (R(g_vp) @ R(w.V2)) * dsilu(k["pre"])                                 # [N,C,H]
    g_A = g_pre.sum(1)
    g_Bc = torch.zeros(B, C, H, dtype=x.dtype).index_add_(0, batch, g_pre)
    G[f"{L}.edge_mlp_virtual.0.weight"][:, 2 * H] += torch.einsum("nch,nc->h", g_pre, k["vr"])
    g_vr = g_pre @ w.w_vr                                                   # [N,C]
    vr = k["vr"]
    inv = torch.where(vr > 0, 1.0 / vr.clamp(min=1e-300), torch.zeros_like(vr))
    g_vd = g_vd + (g_vr * inv).unsqueeze(1) * vd
    g_Z = torch.zeros(B, 3, C, dtype=x.dtype).index_add_(0, batch, g_vd)
    g_x = g_x - g_vd.sum(-1)
    return dict(g_h=g_h, g_x=g_x, g_A=g_A, g_aggm=g_aggm, g_aggx=g_aggx, g_svel=g_svel,
                g_sgrav=g_sgrav, g_Bc=g_Bc, g_Z=g_Z)


# --------------------------------------------------------------------------
# S5 graph_post  (virtual coordinate / feature update)
# --------------------------------------------------------------------------
def graph_post_fwd(w: LayerW, cfg: Config, gptr, Z, HvT, poolV, poolX):
    n_b = (gptr[1:] - gptr[:-1]).clamp(min=1).to(Z.dtype)
    Z_new = Z + poolX / n_b.view(-1, 1, 1)
    pm = poolV / n_b.view(-1, 1, 1)
    z5 = R(HvT) @ R(w.W5a).T + R(pm) @ R(w.W5b).T + w.b5
    out = R(silu(z5)) @ R(w.W6).T + w.b6
    HvT_new = HvT + out if cfg.residual else out
    return Z_new, HvT_new


def graph_post_bwd(w: LayerW, cfg: Config, G, L: str, gptr, HvT, poolV, g_Zn, g_HvTn):
    """returns g_Z, g_HvT, g_poolV, g_poolX."""
    n_b = (gptr[1:] - gptr[:-1]).clamp(min=1).to(HvT.dtype)
    pm = poolV / n_b.view(-1, 1, 1)
    z5 = R(HvT) @ R(w.W5a).T + R(pm) @ R(w.W5b).T + w.b5
    u = silu(z5)
    g_out = g_HvTn
    g_HvT = g_HvTn.clone() if cfg.residual else torch.zeros_like(g_HvTn)
    G[f"{L}.node_mlp_virtual.2.bias"] += g_out.sum((0, 1))
    G[f"{L}.node_mlp_virtual.2.weight"] += torch.einsum("bco,bch->oh", R(g_out), R(u))
    g_z5 = (R(g_out) @ R(w.W6)) * dsilu(z5)
    dW5 = G[f"{L}.node_mlp_virtual.0.weight"]
    G[f"{L}.node_mlp_virtual.0.bias"] += g_z5.sum((0, 1))
    dW5[:, :H] += torch.einsum("bco,bch->oh", R(g_z5), R(HvT))
    dW5[:, H:] += torch.einsum("bco,bch->oh", R(g_z5), R(pm))
    g_HvT = g_HvT + R(g_z5) @ R(w.W5a)
    g_poolV = (R(g_z5) @ R(w.W5b)) / n_b.view(-1, 1, 1)
    g_poolX = g_Zn / n_b.view(-1, 1, 1)
    return g_Zn.clone(), g_HvT, g_poolV, g_poolX


# --------------------------------------------------------------------------
# layer / model composition (what fastegnn_layer_forward / _backward chain)
# --------------------------------------------------------------------------
def layer_forward(p, L, cfg, csr, gptr, batch, ea, h, x, vel, Z, HvT, gravity, node_attr=None):
    w = LayerW(p, L, cfg)
    P, Q, A, svel, sgrav = node_pre_fwd(w, cfg, h)
    xbar, mX, Bc = graph_pre_fwd(w, cfg, x, gptr, Z, HvT)
    aggm, aggx = edge_fwd(w, cfg, csr, P, Q, x, ea)
    h_new, x_new, poolV, poolX = virt_fwd(w, cfg, h, A, Bc, x, vel, Z, batch, aggm, aggx, svel, sgrav,
                                          gravity, node_attr)
    Z_new, HvT_new = graph_post_fwd(w, cfg, gptr, Z, HvT, poolV, poolX)
    saved = dict(h=h, x=x, Z=Z, HvT=HvT, P=P, Q=Q, A=A, Bc=Bc, aggm=aggm, poolV=poolV)
    return h_new, x_new, Z_new, HvT_new, saved


def layer_backward(p, L, cfg, G, csr, gptr, batch, ea, vel, gravity, saved, g_hn, g_xn, g_Zn, g_HvTn,
                   node_attr=None):
    """returns g_h, g_x, g_Z, g_HvT, g_vel (gradients w.r.t. the layer inputs)."""
    w = LayerW(p, L, cfg)
    s = saved
    g_Z, g_HvT, g_poolV, g_poolX = graph_post_bwd(w, cfg, G, L, gptr, s["HvT"], s["poolV"], g_Zn, g_HvTn)
    r = virt_bwd(w, cfg, G, L, s["h"], s["A"], s["Bc"], s["x"], vel, s["Z"], batch, s["aggm"], gravity,
                 g_hn, g_xn, g_poolV, g_poolX, node_attr)
    # svel / sgrav are recomputed by node_pre_bwd; d/dvel = svel * g_xn
    _, _, _, svel, _ = node_pre_fwd(w, cfg, s["h"])
    g_vel = svel.unsqueeze(1) * g_xn
    g_P, g_Q, g_xe = edge_bwd(w, cfg, G, L, csr, s["P"], s["Q"], s["x"], ea, r["g_aggm"], r["g_aggx"])
    g_HvT2, g_Z2, g_xbar = graph_pre_bwd(w, cfg, G, L, s["x"], gptr, s["Z"], s["HvT"], r["g_Bc"])
    g_h = r["g_h"] + node_pre_bwd(w, cfg, G, L, s["h"], g_P, g_Q, r["g_A"], r["g_svel"], r["g_sgrav"])
    g_x = r["g_x"] + g_xe + g_xbar[batch]
    return g_h, g_x, g_Z + r["g_Z"] + g_Z2, g_HvT + g_HvT2, g_vel


def model_forward(p: Params, cfg: Config, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean,
                  edge_attr, node_attr=None):
    with _mode(cfg):
        return _model_forward(p, cfg, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr, node_attr)


def _model_forward(p, cfg, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr, node_attr=None):
    N = node_loc.size(0)
    B = loc_mean.size(0)
    dt = node_loc.dtype
    csr = build_csr(edge_index, N, dt)
    gptr = graph_ptr(data_batch, B)
    ea = edge_attr[csr.perm]
    gravity = torch.tensor(list(cfg.gravity), dtype=dt) if cfg.gravity is not None else None
    HvT = p["virtual_node_feat"][0].T.unsqueeze(0).repeat(B, 1, 1).contiguous()     # [B,C,H]
    h = node_feat @ p["embedding_in.weight"].T + p["embedding_in.bias"]
    x, Z = node_loc, loc_mean
    saved = []
    for i in range(cfg.n_layers):
        h, x, Z, HvT, s = layer_forward(p, f"gcl_{i}", cfg, csr, gptr, data_batch, ea, h, x, node_vel, Z, HvT,
                                        gravity, node_attr)
        saved.append(s)
    ctx = dict(csr=csr, gptr=gptr, ea=ea, gravity=gravity, saved=saved, node_feat=node_feat,
               batch=data_batch, vel=node_vel, node_attr=node_attr)
    return x, Z, ctx


def model_backward(p: Params, cfg: Config, ctx, g_loc, g_vloc):
    """returns (param grads dict, input grads dict)."""
    with _mode(cfg):
        return _model_backward(p, cfg, ctx, g_loc, g_vloc)


def _model_backward(p, cfg, ctx, g_loc, g_vloc):
    G = {k: torch.zeros_like(v) for k, v in p.items()}
    N = g_loc.size(0)
    dt = g_loc.dtype
    B, C = g_vloc.size(0), g_vloc.size(2)
    g_h = torch.zeros(N, H, dtype=dt)
    g_x, g_Z = g_loc, g_vloc
    g_HvT = torch.zeros(B, C, H, dtype=dt)
    g_vel = torch.zeros(N, 3, dtype=dt)
    for i in reversed(range(cfg.n_layers)):
        g_h, g_x, g_Z, g_HvT, gv = layer_backward(p, f"gcl_{i}", cfg, G, ctx["csr"], ctx["gptr"], ctx["batch"],
                                                  ctx["ea"], ctx["vel"], ctx["gravity"], ctx["saved"][i],
                                                  g_h, g_x, g_Z, g_HvT, ctx["node_attr"])
        g_vel = g_vel + gv
    G["virtual_node_feat"] += g_HvT.sum(0).T.unsqueeze(0)
    G["embedding_in.weight"] += g_h.T @ ctx["node_feat"]
    G["embedding_in.bias"] += g_h.sum(0)
    gin = dict(node_feat=g_h @ p["embedding_in.weight"], node_loc=g_x, node_vel=g_vel, loc_mean=g_Z)
    return G, gin
