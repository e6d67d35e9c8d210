"""Drop-in ``EGNN`` baseline (reference ``models/basic.py:285-341``; built at ``main_nbody.py:107`` as
``EGNN(n_layers, in_node_nf=2, in_edge_nf=2, hidden_nf, device, with_v=True)``) on the same HIP kernels as
FastEGNN: it is the virtual-channel-free wiring of the stage kernels (``FASTEGNN_F_EGNN``: C = 0, radial
first in the message MLP's input, coordinate head with bias, +-100 clamp, no residual on h; ``norm=True`` --
the F.normalize of the 1x1 Gram feature, basic.py:271-272 -- is FASTEGNN_F_EGNN_NORM).
Same constructor / forward signature / state_dict keys as the reference class; GPU only."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import List, Optional

import torch
from torch import nn

from . import _lib as K
from .model import RangeGuard, SortedGraph, _PadParams, _PtrTable, _activation_kind, _carve, _fill, _new_layer, _stream

H = K.H


def _base_mlp(i, h, o, act, last_act=False, flat=False):
    if flat:   # BaseMLP(flat=True), basic.py:176-178
        act, h = nn.Tanh(), 4 * h
    mods = [nn.Linear(i, h), act, nn.Linear(h, o)] + ([act] if last_act else [])
    m = nn.Module()
    m.mlp = nn.Sequential(*mods)
    return m


class EGNN_Layer(nn.Module):
    """Parameter holder; construction order of models/basic.py:286-300."""

    def __init__(self, in_edge_nf, hidden_nf, activation, with_v, flat=False):
        super().__init__()
        self.edge_message_net = nn.Module()
        self.edge_message_net.scalar_net = _base_mlp(1 + 2 * hidden_nf + in_edge_nf, hidden_nf, hidden_nf, activation, True, flat)
        self.coord_net = _base_mlp(hidden_nf, hidden_nf, 1, activation, False, flat)
        self.node_net = _base_mlp(2 * hidden_nf, hidden_nf, hidden_nf, activation, False, flat)
        self.node_v_net = _base_mlp(hidden_nf, hidden_nf, 1, activation, False, flat) if with_v else None


_SLOT_OF = {   # FASTEGNN_P_* slot name -> reference key suffix inside layers.<i>
    "edge_mlp.0.weight": "edge_message_net.scalar_net.mlp.0.weight", "edge_mlp.0.bias": "edge_message_net.scalar_net.mlp.0.bias",
    "edge_mlp.2.weight": "edge_message_net.scalar_net.mlp.2.weight", "edge_mlp.2.bias": "edge_message_net.scalar_net.mlp.2.bias",
    "coord_mlp_r.0.weight": "coord_net.mlp.0.weight", "coord_mlp_r.0.bias": "coord_net.mlp.0.bias",
    "coord_mlp_r.2.weight": "coord_net.mlp.2.weight", "coord_mlp_r.2.bias": "coord_net.mlp.2.bias",
    "node_mlp.0.weight": "node_net.mlp.0.weight", "node_mlp.0.bias": "node_net.mlp.0.bias",
    "node_mlp.2.weight": "node_net.mlp.2.weight", "node_mlp.2.bias": "node_net.mlp.2.bias",
    "coord_mlp_vel.0.weight": "node_v_net.mlp.0.weight", "coord_mlp_vel.0.bias": "node_v_net.mlp.0.bias",
    "coord_mlp_vel.2.weight": "node_v_net.mlp.2.weight", "coord_mlp_vel.2.bias": "node_v_net.mlp.2.bias",
}


def _egnn_pad_layout(name: str, shape, h: int):
    """64-wide image of a parameter of a ``hidden_nf = h < 64`` EGNN (same algebra as fastegnn_amd.model._pad_layout: zero
    rows give zero pre-activations, SiLU(0) = 0, zero columns ignore their input).  Input blocks per models/basic.py:
    message MLP on [radial | h_row | h_col | edge_fea] (:313, one leading non-hidden column), node_net on [h | agg] (:317),
    the other first layers on one hidden vector; every first / second layer but the [1,H] heads has hidden-sized outputs."""
    shape = tuple(shape)
    mod, kind = name.rsplit(".", 1)
    head = mod.endswith(("coord_net.mlp.2", "node_v_net.mlp.2"))           # [1, H] -> scalar
    if kind == "bias":
        return (1, shape[0], 1, [], shape) if head else (1, shape[0], 1, [h], (H,))
    rows, cols = shape
    if mod == "embedding":
        blocks, lead = [], 0
    elif mod.endswith("scalar_net.mlp.0"):
        blocks, lead = [h, h], 1
    elif mod.endswith("node_net.mlp.0"):
        blocks, lead = [h, h], 0
    else:
        blocks, lead = [h], 0
    rows_dst = rows if head else H
    return rows, cols, rows_dst, blocks, (rows_dst, cols + sum(b // h * (H - h) for b in blocks)), lead


class _EGNNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, graph, edge_fea, x, h_in, v, *params):
        lib = K.lib(act=spec.act_kind != K.ACT_SILU, wide=spec.wide)
        ctx.wide = spec.wide
        ctx.guard = getattr(spec, "guard", None)   # the module's RangeGuard while it runs on the f16x2 build (fastegnn_amd.model)
        # edge_fea is a differentiable input (basic.py:313 concatenates it into the message MLP's input): its gradient is
        # accumulated by the edge backward kernel in sorted-edge order when asked for
        ea_sorted = graph.permute(edge_fea.detach() if edge_fea is not None else None)
        dev = x.device
        st = _stream(dev)
        N = x.size(0)
        f32 = dict(dtype=torch.float32, device=dev)
        params = [p.detach() for p in params]
        x, h_in, v = x.detach().contiguous().float(), h_in.detach().contiguous().float(), v.detach().contiguous().float()
        batch = torch.zeros(N, dtype=torch.int32, device=dev)
        h = torch.empty(N, H, **f32)
        K.check(lib.fastegnn_embed_forward(K.ptr(h_in), N, spec.nf, K.ptr(params[0]), K.ptr(params[1]), K.ptr(h), st),
                "fastegnn_embed_forward")
        nwp = lib.fastegnn_wpack_floats(0)
        saved = []
        for i in range(spec.n_layers):
            tab = _PtrTable([params[s] if s is not None else None for s in spec.layer_slots[i]])
            b = dict(h=h, x=x)
            b.update(_carve(dev, dict(wpack=(nwp,), P=(N, H), QX=(N, K.QX_LD), A=(N, H), svel=(N,), aggm=(N, H),
                                      aggx=(N, 3), npre=(N, H))))
            b.update(h_out=torch.empty(N, H, **f32), x_out=torch.empty(N, 3, **f32))
            L = _new_layer(spec, N, 1, graph)
            _fill(L, batch=batch, vel=v, params=tab.addr(), **b)
            L.QX_src = b["QX"].data_ptr()
            if ea_sorted is not None:
                L.ea_sorted = ea_sorted.data_ptr()
            K.check(lib.fastegnn_layer_forward(C.byref(L), st), f"fastegnn_layer_forward[egnn {i}]")
            saved.append(b)
            h, x = b["h_out"], b["x_out"]
            del b["h_out"], b["x_out"]
        ctx.spec, ctx.graph, ctx.saved = spec, graph, saved
        ctx.misc = (batch, ea_sorted, h_in, v, params)
        return x, h

    @staticmethod
    def backward(ctx, g_x, g_h):
        spec, graph, saved = ctx.spec, ctx.graph, ctx.saved
        lib = K.lib(act=spec.act_kind != K.ACT_SILU, wide=ctx.wide)
        batch, ea_sorted, h_in, v, params = ctx.misc
        dev = v.device
        st = _stream(dev)
        N, E = v.size(0), graph.E
        f32 = dict(dtype=torch.float32, device=dev)
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]
        flat = torch.zeros(sum(sizes), **f32)
        grads, off = [], 0
        for p, s in zip(params, sizes):
            grads.append(flat[off:off + p.numel()].view_as(p))
            off += s
        g_h = (g_h if g_h is not None else torch.zeros(N, H, **f32)).contiguous().float()
        g_x = (g_x if g_x is not None else torch.zeros(N, 3, **f32)).contiguous().float()
        g_vel = torch.zeros(N, 3, **f32)
        sc = _carve(dev, dict(g_A=(N, H), g_P=(N, H), g_aggm=(N, H), g_aggx=(N, 3), g_svel=(N,),
                              g_QXe=(max(E, 1) if spec.flags & K.F_DETERMINISTIC else 1, K.QX_LD), g_QX_src=(N, K.QX_LD), g_xrow=(N, 3),
                              wg_edge=(lib.fastegnn_wg_edge_floats(E),), wg_node=(lib.fastegnn_wg_node_floats(N, 1, 0),), wg_slab=(lib.fastegnn_wg_slab_floats(),)))
        sc["g_xbar"] = torch.zeros(1, 4, **f32)
        want_ea = ctx.needs_input_grad[2] and ea_sorted is not None and E > 0
        g_ea_sorted = torch.zeros(E, spec.ea, **f32) if want_ea else None
        for i in reversed(range(spec.n_layers)):
            b = saved[i]
            ptab = _PtrTable([params[s] if s is not None else None for s in spec.layer_slots[i]])
            gtab = _PtrTable([grads[s] if s is not None else None for s in spec.layer_slots[i]])
            out = dict(g_h=torch.empty(N, H, **f32), g_x=torch.empty(N, 3, **f32))
            L = _new_layer(spec, N, 1, graph)
            _fill(L, batch=batch, vel=v, params=ptab.addr(), grads=gtab.addr(), g_h_out=g_h, g_x_out=g_x, g_vel=g_vel,
                  **b, **out, **sc)
            L.QX_src = b["QX"].data_ptr()
            L.g_QX = sc["g_QX_src"].data_ptr()
            if ea_sorted is not None:
                L.ea_sorted = ea_sorted.data_ptr()
            if g_ea_sorted is not None:
                L.g_ea_sorted = g_ea_sorted.data_ptr()
            K.check(lib.fastegnn_layer_backward(C.byref(L), st), f"fastegnn_layer_backward[egnn {i}]")
            g_h, g_x = out["g_h"], out["g_x"]
            saved[i] = None
        g_hin = torch.empty_like(h_in) if ctx.needs_input_grad[4] else None
        K.check(lib.fastegnn_embed_backward(K.ptr(h_in), K.ptr(g_h), N, spec.nf, K.ptr(params[0]), K.ptr(grads[0]),
                                            K.ptr(grads[1]), K.ptr(g_hin), st), "fastegnn_embed_backward")
        g_ea = None
        if g_ea_sorted is not None:   # back to the caller's edge order: sorted edge k is input edge perm[k]
            g_ea = torch.empty_like(g_ea_sorted)
            g_ea.index_copy_(0, graph.perm[:E].long(), g_ea_sorted)
        if ctx.guard is not None:   # a forward that left the fp16 operand range hands ZERO parameter gradients on (model.RangeGuard)
            ctx.guard.zero_if_flagged(lib, flat)
        return (None, None, g_ea, g_x, g_hin, g_vel, *grads)


class EGNN(nn.Module):
    """MI355X-native drop-in for the reference ``EGNN`` (models/basic.py:323-341)."""

    def __init__(self, n_layers, in_node_nf, in_edge_nf, hidden_nf, activation=nn.SiLU(), device='cpu', with_v=False,
                 flat=False, norm=False):
        super().__init__()
        # hidden_nf <= 64 with flat=False: the fused kernels (FASTEGNN_F_EGNN wiring).  flat=True (Tanh MLPs with 4 x hidden inner
        # units, basic.py:176-178) or 64 < hidden_nf <= 256: the unfused wide path (fastegnn_amd/wide.py: generic-width operators)
        if not 1 <= hidden_nf <= 256:
            raise NotImplementedError(f"fastegnn_amd.EGNN: hidden_nf must be at most 256 in this build (got {hidden_nf})")
        self._wide = bool(flat) or hidden_nf > H
        self.flat = bool(flat)
        # `activation` (basic.py:324; BaseMLP puts it behind every first layer and behind the message MLP, :181-192): the same
        # kinds as FastEGNN's act_fn, on the generic-activation build of the library when it is not SiLU
        self._act = _activation_kind(activation)
        if self._act[0] in (K.ACT_SIGMOID, K.ACT_SOFTPLUS) and hidden_nf < H and not self._wide:
            raise NotImplementedError("fastegnn_amd.EGNN: hidden_nf < 64 runs zero-padded, which needs activation(0) = 0")
        self.hidden_nf = hidden_nf
        self.norm = bool(norm)
        if not self._wide and (in_edge_nf > 7 or in_node_nf > 8):
            raise NotImplementedError("fastegnn_amd.EGNN: in_edge_nf<=7, in_node_nf<=8")
        self.n_layers, self.with_v = n_layers, with_v
        self.in_node_nf, self.in_edge_nf = in_node_nf, in_edge_nf
        self.layers = nn.ModuleList()                      # registered first, filled after the embedding (basic.py:327-335)
        self.embedding = nn.Linear(in_node_nf, hidden_nf)
        for _ in range(n_layers):
            self.layers.append(EGNN_Layer(in_edge_nf, hidden_nf, activation, with_v, flat))
        self._spec = None
        self._graph_cache = {}
        self._range = RangeGuard()   # automatic wide-range fallback (fastegnn_amd.model.RangeGuard)
        self.deterministic = bool(K.deterministic_default())   # see fastegnn_amd.FastEGNN.deterministic (set before the first call)
        self.to(device)

    def _build_spec(self):
        pidx = dict(self.named_parameters())
        names = ["embedding.weight", "embedding.bias"]
        layer_slots: List[List[Optional[int]]] = []
        for i in range(self.n_layers):
            slots = []
            for slot in K.PARAM_SLOTS:
                key = f"layers.{i}.{_SLOT_OF[slot]}" if slot in _SLOT_OF else None
                if key is not None and key in pidx:
                    slots.append(len(names))
                    names.append(key)
                else:
                    slots.append(None)
            layer_slots.append(slots)
        self._plist = [pidx[n] for n in names]
        self._spec = SimpleNamespace(C=0, ea=self.in_edge_nf, na=0, nf=self.in_node_nf, n_layers=self.n_layers,
                                     flags=K.F_EGNN | (K.F_EGNN_NORM if self.norm else 0) | (K.F_DETERMINISTIC if self.deterministic else 0)
                                     | (self._act[0] << K.F_ACT_SHIFT),
                                     act_kind=self._act[0], act_param=self._act[1], gravity=[0.0, 0.0, 0.0],
                                     layer_slots=layer_slots, names=names, wide=False)

    def forward(self, x, h, edge_index, edge_fea, v=None):
        if not x.is_cuda:
            raise RuntimeError("fastegnn_amd.EGNN runs on a gfx950 GPU only (no CPU fallback)")
        if self._wide:
            from . import wide
            from .model import _DEBUG_CHECKS, _check_indices
            if _DEBUG_CHECKS:   # FASTEGNN_DEBUG_CHECKS=1: an out-of-range row / col would be an out-of-bounds gather or atomic
                _check_indices(edge_index, torch.zeros(x.size(0), dtype=torch.long, device=x.device), x.size(0), 1)
            if edge_fea is not None and edge_fea.size(1) == 0:
                edge_fea = None
            x_out, h_out = wide.egnn_forward(self, x, h, edge_index, edge_fea, v)
            return (x_out, v, h_out) if v is not None else (x_out, h_out)
        if self._spec is None:
            self._build_spec()
        N = x.size(0)
        key = (edge_index.data_ptr(), edge_index.size(1), edge_index._version, N)
        graph = self._graph_cache.get(key)
        if graph is None:
            graph = SortedGraph(edge_index, N, csc=bool(self._spec.flags & K.F_DETERMINISTIC))
            graph._keepalive = edge_index
            if len(self._graph_cache) >= 8:
                self._graph_cache.pop(next(iter(self._graph_cache)))
            self._graph_cache[key] = graph
        if edge_fea is not None and edge_fea.size(1) == 0:
            edge_fea = None
        vv = v if v is not None else torch.zeros_like(x)
        plist = self._plist
        if self.hidden_nf < H:   # 64-wide images of the parameters (fastegnn_pad_params; the reverse mode slices the gradients back)
            plist = list(_PadParams.apply(tuple(self._spec.names), self.hidden_nf, 0, _egnn_pad_layout, *plist))
        guard, spec = self._range, self._spec
        if not guard.wide:
            guard.poll("EGNN", self._plist)      # no synchronisation: an overflow of an EARLIER pass switches the build here (model.RangeGuard)
        spec.wide = guard.wide
        spec.guard = None if guard.wide else guard
        x_out, h_out = _EGNNFunction.apply(spec, graph, edge_fea, x, h, vv, *plist)
        if not guard.wide:
            guard.launch(K.lib(act=spec.act_kind != K.ACT_SILU, wide=False), (x_out, h_out), (x, h))
            if guard.mode == "sync" and guard.sync_and_poll(x.device, "EGNN", self._plist):
                spec.wide, spec.guard = True, None
                x_out, h_out = _EGNNFunction.apply(spec, graph, edge_fea, x, h, vv, *plist)
        if self.hidden_nf < H:
            h_out = h_out[:, :self.hidden_nf]     # the padded features are identically zero
        return (x_out, v, h_out) if v is not None else (x_out, h_out)
