"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the FastEGNN forward/backward hot path.

Plain-PyTorch, *unfused*, op-for-op restatement of the reference algorithm
(gathers, concatenations, dense layers, scatter-adds, mean pools), written as
pure functions over a flat parameter dictionary whose keys and shapes are the
reference's ``state_dict`` (``/root/reference/models/FastEGNN.py:227-263``).
Backward comes from autograd, like the reference (``utils/train.py:169``).

Parity pinning: checked in ``tests/test_oracle_golden.py`` against the golden
vectors under ``tests/golden/`` that ``oracle/gen_goldens.py`` produced by
importing the real reference in the build container.

Nothing under ``fastegnn_amd/`` imports this file; it is only the checker
(tests, ``__graft_entry__.smoke()``) and the timed ``cpu_baseline`` of
``bench.py``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass
class Config:
    """Constructor arguments of the reference model (FastEGNN.py:227-228)."""
    node_feat_nf: int
    node_attr_nf: int
    edge_attr_nf: int
    hidden_nf: int
    virtual_channels: int
    n_layers: int = 4
    residual: bool = True
    attention: bool = False
    normalize: bool = False
    tanh: bool = False
    gravity: Optional[Sequence[float]] = None
    coords_agg: str = "mean"          # E_GCL_vel default, FastEGNN.py:12
    bf16: bool = False                # bf16 operand mode of the product (BASELINE configs[2]); honoured by
                                      # oracle/factored.py only -- the reference itself has no reduced-precision mode
    epsilon: float = 1e-8             # FastEGNN.py:21
    act: str = "silu"                 # the constructor's act_fn (FastEGNN.py:227): silu | relu | leaky_relu | tanh | sigmoid | elu |
    act_param: float = 0.0            # gelu | softplus; negative_slope / alpha / beta where the module has one


# --------------------------------------------------------------------------
# parameter construction (same distributions as nn.Linear / xavier gain 1e-3)
# --------------------------------------------------------------------------
def _linear(gen, out_f, in_f, bias=True, dtype=torch.float32):
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen, dtype=torch.float64) * 2 - 1) * bound
    out = {"weight": w.to(dtype)}
    if bias:
        b = (torch.rand(out_f, generator=gen, dtype=torch.float64) * 2 - 1) * bound
        out["bias"] = b.to(dtype)
    return out


def _xavier(gen, out_f, in_f, gain, dtype=torch.float32):
    bound = gain * math.sqrt(6.0 / (in_f + out_f))
    w = (torch.rand(out_f, in_f, generator=gen, dtype=torch.float64) * 2 - 1) * bound
    return {"weight": w.to(dtype)}


def init_params(cfg: Config, seed: int = 0, coord_gain: float = 1e-3,
                dtype=torch.float32) -> Params:
    """Random parameters with the reference's names/shapes (FastEGNN.py:28-99,256-257).

    ``coord_gain`` scales the three gain-1e-3 coordinate heads; tests use a
    "trained-like" variant (gain 0.1) so the coordinate paths matter.
    """
    g = torch.Generator().manual_seed(seed)
    H, C = cfg.hidden_nf, cfg.virtual_channels
    p: Params = {}

    def put(prefix, d):
        for k, v in d.items():
            p[f"{prefix}.{k}"] = v

    p["virtual_node_feat"] = torch.randn(1, H, C, generator=g, dtype=torch.float64).to(dtype)
    put("embedding_in", _linear(g, H, cfg.node_feat_nf, dtype=dtype))
    for i in range(cfg.n_layers):
        L = f"gcl_{i}"
        put(f"{L}.edge_mlp.0", _linear(g, H, 2 * H + 1 + cfg.edge_attr_nf, dtype=dtype))
        put(f"{L}.edge_mlp.2", _linear(g, H, H, dtype=dtype))
        put(f"{L}.edge_mlp_virtual.0", _linear(g, H, 2 * H + 1 + C, dtype=dtype))
        put(f"{L}.edge_mlp_virtual.2", _linear(g, H, H, dtype=dtype))
        if cfg.attention:
            put(f"{L}.att_mlp.0", _linear(g, 1, H, dtype=dtype))
            put(f"{L}.att_mlp_virtual.0", _linear(g, 1, H, dtype=dtype))
        for name in ("coord_mlp_r", "coord_mlp_r_virtual", "coord_mlp_v_virtual"):
            put(f"{L}.{name}.0", _linear(g, H, H, dtype=dtype))
            put(f"{L}.{name}.2", _xavier(g, 1, H, coord_gain, dtype=dtype))
        put(f"{L}.coord_mlp_vel.0", _linear(g, H, H, dtype=dtype))
        put(f"{L}.coord_mlp_vel.2", _linear(g, 1, H, dtype=dtype))
        if cfg.gravity is not None:
            put(f"{L}.gravity_mlp.0", _linear(g, H, H, dtype=dtype))
            put(f"{L}.gravity_mlp.2", _linear(g, 1, H, dtype=dtype))
        put(f"{L}.node_mlp.0", _linear(g, H, 2 * H + C * H + cfg.node_attr_nf, dtype=dtype))
        put(f"{L}.node_mlp.2", _linear(g, H, H, dtype=dtype))
        put(f"{L}.node_mlp_virtual.0", _linear(g, H, 2 * H, dtype=dtype))
        put(f"{L}.node_mlp_virtual.2", _linear(g, H, H, dtype=dtype))
    return p


# --------------------------------------------------------------------------
# helpers restating the reference's segment ops
# --------------------------------------------------------------------------
def segment_sum(data, ids, n):
    """FastEGNN.py:279-284 -- zero buffer + scatter_add_ with an expanded index."""
    out = data.new_zeros((n, data.size(1)))
    out.scatter_add_(0, ids.unsqueeze(-1).expand(-1, data.size(1)), data)
    return out


def segment_mean(data, ids, n):
    """FastEGNN.py:287-294 -- sum / count.clamp(min=1)."""
    idx = ids.unsqueeze(-1).expand(-1, data.size(1))
    s = data.new_zeros((n, data.size(1)))
    c = data.new_zeros((n, data.size(1)))
    s.scatter_add_(0, idx, data)
    c.scatter_add_(0, idx, torch.ones_like(data))
    return s / c.clamp(min=1)


def graph_mean_pool(x, batch, n_graphs=None):
    """torch_geometric 2.5.2 ``global_mean_pool`` (requirements.txt:14): per-graph
    mean, graph count = batch.max()+1, empty graphs give 0 (count clamped to 1).
    Call sites: FastEGNN.py:148,170,212."""
    if n_graphs is None:
        n_graphs = int(batch.max()) + 1
    s = x.new_zeros((n_graphs, x.size(1)))
    s.index_add_(0, batch, x)
    c = x.new_zeros(n_graphs)
    c.index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return s / c.clamp(min=1).unsqueeze(-1)


def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def act_of(cfg):
    """The activation module of the reference constructor as a function (default nn.SiLU(), FastEGNN.py:227)."""
    a, q = getattr(cfg, "act", "silu"), getattr(cfg, "act_param", 0.0)
    return {"silu": F.silu, "relu": F.relu, "leaky_relu": lambda z: F.leaky_relu(z, q), "tanh": torch.tanh,
            "sigmoid": torch.sigmoid, "elu": lambda z: F.elu(z, q), "gelu": F.gelu,
            "softplus": lambda z: F.softplus(z, beta=q, threshold=20.0)}[a]


def _mlp2(p, name, x, act_last, act=F.silu):
    """Linear -> act -> Linear [-> act]."""
    y = _lin(p, name + ".2", act(_lin(p, name + ".0", x)))
    return act(y) if act_last else y


def _coord_head(p, name, x, tanh, act=F.silu):
    y = _mlp2(p, name, x, act_last=False, act=act)
    return torch.tanh(y) if tanh else y


# --------------------------------------------------------------------------
# one E_GCL_vel layer, op for op (FastEGNN.py:192-223)
# --------------------------------------------------------------------------
def layer_forward(p: Params, L: str, cfg: Config, h, edge_index, x, vel, Z, Hv, batch,
                  edge_attr=None, node_attr=None, gravity=None):
    row, col = edge_index[0], edge_index[1]
    act = act_of(cfg)
    N, C, H = h.size(0), cfg.virtual_channels, cfg.hidden_nf

    # coord2radial, :180-189
    d = x[row] - x[col]
    radial = (d ** 2).sum(1, keepdim=True)
    if cfg.normalize:
        d = d / (radial.sqrt().detach() + cfg.epsilon)

    # virtual geometry, :206-207
    vd = Z[batch] - x.unsqueeze(-1)                       # [N,3,C]
    vr = torch.norm(vd, p=2, dim=1, keepdim=True)         # [N,1,C]

    # edge_model, :102-108
    m = _mlp2(p, f"{L}.edge_mlp", torch.cat([h[row], h[col], radial, edge_attr], dim=1), act_last=True, act=act)
    if cfg.attention:
        m = m * torch.sigmoid(_lin(p, f"{L}.att_mlp.0", m))

    # centroid + Gram, :212-214
    xbar = graph_mean_pool(x, batch)
    mz = Z - xbar.unsqueeze(-1)
    mX = torch.einsum("bij,bjk->bik", mz.permute(0, 2, 1), mz)   # [B,C,C]

    # edge_mode_virtual, :111-119
    inp = torch.cat([h.unsqueeze(-1).repeat(1, 1, C), Hv[batch], vr, mX[batch]], dim=1)  # [N,2H+1+C,C]
    v = _mlp2(p, f"{L}.edge_mlp_virtual", inp.permute(0, 2, 1), act_last=True, act=act)           # [N,C,H]
    if cfg.attention:
        v = v * torch.sigmoid(_lin(p, f"{L}.att_mlp_virtual.0", v))
    v_hc = v.permute(0, 2, 1)                                                             # [N,H,C]

    # coord_model_vel, :122-144
    trans = d * _coord_head(p, f"{L}.coord_mlp_r", m, cfg.tanh, act)
    if cfg.coords_agg == "sum":
        agg = segment_sum(trans, row, N)
    elif cfg.coords_agg == "mean":
        agg = segment_mean(trans, row, N)
    else:
        raise Exception("Wrong coords_agg parameter")
    x_new = x + agg
    phi_xv = _coord_head(p, f"{L}.coord_mlp_r_virtual", v, cfg.tanh, act).permute(0, 2, 1)     # [N,1,C]
    x_new = x_new + torch.mean(-vd * phi_xv, dim=-1)
    x_new = x_new + _mlp2(p, f"{L}.coord_mlp_vel", h, act_last=False, act=act) * vel
    if gravity is not None:
        x_new = x_new + _mlp2(p, f"{L}.gravity_mlp", h, act_last=False, act=act) * gravity

    # coord_model_virtual, :146-150
    phi_X = _coord_head(p, f"{L}.coord_mlp_v_virtual", v, cfg.tanh, act).permute(0, 2, 1)      # [N,1,C]
    Z_new = Z + graph_mean_pool((vd * phi_X).reshape(N, -1), batch).reshape(-1, 3, C)

    # node_model, :153-166
    agg_m = segment_mean(m, row, N)
    parts = [h, agg_m, v_hc.reshape(N, -1)]
    if node_attr is not None:
        parts.append(node_attr)
    out = _mlp2(p, f"{L}.node_mlp", torch.cat(parts, dim=1), act_last=False, act=act)
    h_new = h + out if cfg.residual else out

    # node_model_virtual, :168-177
    pool = graph_mean_pool(v_hc.reshape(N, -1), batch).reshape(-1, H, C)
    outv = _mlp2(p, f"{L}.node_mlp_virtual", torch.cat([Hv, pool], dim=1).permute(0, 2, 1),
                 act_last=False, act=act).permute(0, 2, 1)
    Hv_new = Hv + outv if cfg.residual else outv
    return h_new, x_new, Hv_new, Z_new


def forward(p: Params, cfg: Config, node_feat, node_loc, node_vel, edge_index, data_batch,
            loc_mean, edge_attr=None, node_attr=None, return_layers=False):
    """FastEGNN.forward, FastEGNN.py:265-276 -> (node_loc [N,3], virtual_node_loc [B,3,C])."""
    B = int(data_batch[-1]) + 1
    Hv = p["virtual_node_feat"].repeat(B, 1, 1)
    Z = loc_mean
    gravity = None
    if cfg.gravity is not None:
        # reference builds an int64 tensor for [0,-1,0] (FastEGNN.py:258-259) and
        # multiplies it into fp32; the product promotes to the float dtype.
        gravity = torch.tensor(list(cfg.gravity), dtype=node_loc.dtype)
    h = _lin(p, "embedding_in", node_feat)
    x = node_loc
    layers = []
    for i in range(cfg.n_layers):
        h, x, Hv, Z = layer_forward(p, f"gcl_{i}", cfg, h, edge_index, x, node_vel, Z, Hv,
                                    data_batch, edge_attr=edge_attr, node_attr=node_attr,
                                    gravity=gravity)
        if return_layers:
            layers.append((h, x, Hv, Z))
    if return_layers:
        return x, Z, layers
    return x, Z


# --------------------------------------------------------------------------
# caller-side pieces restated for the training-step fixtures (utils/train.py)
# --------------------------------------------------------------------------
def augment_edge_attr(edge_attr, loc_0, edge_index):
    """utils/train.py:41-43 -- append the recomputed edge length."""
    row, col = edge_index[0], edge_index[1]
    length = ((loc_0[row] - loc_0[col]) ** 2).sum(1).sqrt().unsqueeze(1)
    return torch.cat([edge_attr, length], dim=1)


def mmd_kernel(x, y, sigma):
    """utils/train.py:17-20."""
    return torch.exp(-torch.cdist(x, y, p=2) / (2 * sigma * sigma))


def loss_mse_mmd(loc_pred, vloc, loc_t, sample_idx, sigma, weight):
    """utils/train.py:104-165, equal-sized-graph branch (:144-160) with the sampled
    real-node indices passed in explicitly (the reference draws torch.randperm)."""
    mse = F.mse_loss(loc_pred, loc_t)
    V = vloc.permute(0, 2, 1)                      # [B,C,3]
    B, C, _ = V.shape
    R = loc_pred.reshape(B, -1, 3)[:, sample_idx, :]
    l_vv = mmd_kernel(V, V, sigma).sum() / B / C / C
    l_rv = 2 * mmd_kernel(R, V, sigma).sum() / B / R.size(1) / C
    return mse + weight * (l_vv - l_rv), mse


def loss_mse_mmd_nodes(loc_pred, vloc, loc_t, sample_nodes, sigma, weight):
    """Both branches of utils/train.py:104-165 in one form: `sample_nodes` [B,S] are the absolute
    indices of the sampled real nodes of each graph (equal-sized branch :144-160: b*n + idx[s];
    'Simulation' branch :118-142: per-graph draws).  Returns (loss, mse)."""
    mse = F.mse_loss(loc_pred, loc_t)
    V = vloc.permute(0, 2, 1)                      # [B,C,3]
    B, C, _ = V.shape
    S = sample_nodes.size(1)
    R = loc_pred[sample_nodes.reshape(-1)].reshape(B, S, 3)
    l_vv = mmd_kernel(V, V, sigma).sum() / B / C / C
    l_rv = 2 * mmd_kernel(R, V, sigma).sum() / B / S / C
    return mse + weight * (l_vv - l_rv), mse


def adam_step(params, grads, state, step, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12):
    """torch.optim.Adam (non-amsgrad, L2 weight decay folded into the gradient), as constructed at
    main_nbody.py:137.  `state` maps name -> (exp_avg, exp_avg_sq); updated in place."""
    b1, b2 = betas
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        g = g + weight_decay * p
        m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
