"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the EGNN baseline:
op-for-op restatement of ``EGNN`` / ``EGNN_Layer`` / ``InvariantScalarNet`` / ``BaseMLP`` /
``aggregate`` of ``/root/reference/models/basic.py:27-52,172-196,245-341`` (``flat=False``, ``norm`` optional),
over the reference's state_dict keys.  Pinned by tests/golden/egnn_*.npz (oracle/gen_goldens.py --egnn)."""
import torch
import torch.nn.functional as F


def _aggregate_mean(message, row, n):
    """basic.py:27-52 with aggr='mean': scatter_add / count.clamp(min=1)."""
    idx = row.unsqueeze(-1).expand(-1, message.size(1))
    s = message.new_zeros((n, message.size(1))).scatter_add_(0, idx, message)
    c = message.new_zeros((n, message.size(1))).scatter_add_(0, idx, torch.ones_like(message))
    return s / c.clamp(min=1)


def _mlp(p, name, x, last_act, act=F.silu):
    """BaseMLP (basic.py:172-196, flat=False): Linear -> activation -> Linear [-> activation]"""
    y = F.linear(act(F.linear(x, p[name + ".mlp.0.weight"], p[name + ".mlp.0.bias"])),
                 p[name + ".mlp.2.weight"], p[name + ".mlp.2.bias"])
    return act(y) if last_act else y


def layer_forward(p, L, x, h, edge_index, edge_fea, v=None, norm=False, act=F.silu):
    """EGNN_Layer.forward, basic.py:302-320; `act`: the constructor's `activation` (basic.py:280, default nn.SiLU())."""
    row, col = edge_index[0], edge_index[1]
    rij = x[row] - x[col]
    scalar = (rij * rij).sum(1, keepdim=True)                 # 1x1 Gram of the single vector, :275
    if norm:
        scalar = F.normalize(scalar, p=2, dim=-1)
    message = _mlp(p, f"{L}.edge_message_net.scalar_net", torch.cat((scalar, h[row], h[col], edge_fea), -1), True, act)
    f = rij * _mlp(p, f"{L}.coord_net", message, False, act)
    tot_f = torch.clamp(_aggregate_mean(f, row, x.size(0)), min=-100, max=100)
    if v is not None:
        x = x + _mlp(p, f"{L}.node_v_net", h, False, act) * v + tot_f
    else:
        x = x + tot_f
    h = _mlp(p, f"{L}.node_net", torch.cat((h, _aggregate_mean(message, row, x.size(0))), -1), False, act)
    return x, h


def forward(p, n_layers, x, h, edge_index, edge_fea, v=None, norm=False, act=F.silu):
    """EGNN.forward, basic.py:337-341 -> (x, h)."""
    h = F.linear(h, p["embedding.weight"], p["embedding.bias"])
    for i in range(n_layers):
        x, h = layer_forward(p, f"layers.{i}", x, h, edge_index, edge_fea, v, norm, act)
    return x, h
