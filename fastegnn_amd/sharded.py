"""One large graph sharded over the ranks of a node ("row-owner" scheme).

Rank r owns the contiguous node range [r*Npad, (r+1)*Npad) and every edge whose aggregation row lies
in it, so all segment means are complete locally.  Per layer the ranks exchange exactly:

  forward   all-gather   QX  [Npad,68] -> QX_src [W*Npad,68]   (the gathered source table)
            all-reduce   xsum [B,4]                            (centroids / node counts)
            all-reduce   poolV|poolX [B,C,64]+[B,3,C]          (virtual-node accumulators)
  backward  all-reduce   g_Bc|g_Zp                             (adjoint of the broadcast virtual state)
            reduce-scatter g_QX_src [W*Npad,68] -> g_QX [Npad,68]  (transpose of the all-gather)
  once      all-reduce   parameter gradients (fastegnn_amd.dist.allreduce_gradients, by the caller)

The virtual state (Z, Hv) is replicated and updated identically everywhere; the per-graph stages
(graph_pre / graph_post) run redundantly on every rank and only rank 0 keeps their weight gradients.
The compute of every stage is one C-ABI call (include/fastegnn_hip.h); this file is only the
orchestration, written against a small backend interface so that the world_size-2 gloo test can drive
it on CPU with the oracle's stage functions (tests/test_sharded_cpu.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import _lib as K
from .model import FastEGNN, SortedGraph, _PtrTable, _Spec, _carve, _fill, _new_layer, _stream

H = K.H


# ------------------------------------------------------------------------------------------
# backend: everything that touches device memory / kernels
# ------------------------------------------------------------------------------------------
class HipBackend:
    """Stage calls on libfastegnn_hip.so (the product path)."""

    def __init__(self, device):
        self.dev = device
        self.lib = K.lib()

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.dev)

    def carve(self, shapes):
        return _carve(self.dev, shapes)

    def wpack_floats(self, Cn):
        return self.lib.fastegnn_wpack_floats(Cn)

    def wg_slab_floats(self):
        return self.lib.fastegnn_wg_slab_floats()

    def wg_edge_floats(self, E):
        return self.lib.fastegnn_wg_edge_floats(E)

    def build_graph(self, edge_index, n_rows, n_src, row_begin):
        return SortedGraph(edge_index, n_rows, n_src, row_begin)

    def build_batch(self, data_batch, N, B):
        b32 = torch.empty(N, dtype=torch.int32, device=self.dev)
        gptr = torch.empty(B + 1, dtype=torch.int32, device=self.dev)
        K.check(self.lib.fastegnn_build_batch(K.ptr(data_batch.contiguous()), N, B, K.ptr(b32), K.ptr(gptr),
                                              _stream(self.dev)), "fastegnn_build_batch")
        return b32, gptr

    def embed_forward(self, node_feat, nf, W, b, h):
        K.check(self.lib.fastegnn_embed_forward(K.ptr(node_feat), node_feat.size(0), nf, K.ptr(W), K.ptr(b), K.ptr(h),
                                                _stream(self.dev)), "fastegnn_embed_forward")

    def embed_backward(self, node_feat, g_h, nf, W, gW, gb, g_nf):
        K.check(self.lib.fastegnn_embed_backward(K.ptr(node_feat), K.ptr(g_h), node_feat.size(0), nf, K.ptr(W),
                                                 K.ptr(gW), K.ptr(gb), K.ptr(g_nf), _stream(self.dev)),
                "fastegnn_embed_backward")

    def virtual_init(self, vnf, B, Cn, HvT):
        K.check(self.lib.fastegnn_virtual_init(K.ptr(vnf), B, Cn, K.ptr(HvT), _stream(self.dev)), "fastegnn_virtual_init")

    def virtual_init_backward(self, g_HvT, B, Cn, g_vnf):
        K.check(self.lib.fastegnn_virtual_init_backward(K.ptr(g_HvT), B, Cn, K.ptr(g_vnf), _stream(self.dev)),
                "fastegnn_virtual_init_backward")

    def stage(self, name, spec, N, B, graph, t: Dict[str, torch.Tensor], params, grads=None):
        L = _new_layer(spec, N, B, graph)
        ptab = _PtrTable(params)
        keep = [ptab]
        _fill(L, params=ptab.addr(), **{k: v for k, v in t.items() if v is not None})
        if grads is not None:
            gtab = _PtrTable(grads)
            keep.append(gtab)
            _fill(L, grads=gtab.addr())
        K.check(getattr(self.lib, "fastegnn_" + name)(C.byref(L), _stream(self.dev)), "fastegnn_" + name)


# ------------------------------------------------------------------------------------------
# partition
# ------------------------------------------------------------------------------------------
class ShardPlan:
    """Contiguous node ranges of equal padded size Npad = ceil(N / W)."""

    def __init__(self, n_nodes: int, world: int, rank: int):
        self.N, self.world, self.rank = n_nodes, world, rank
        self.Npad = (n_nodes + world - 1) // world
        self.n0 = min(n_nodes, rank * self.Npad)
        self.n1 = min(n_nodes, self.n0 + self.Npad)
        self.nloc = self.n1 - self.n0
        if self.nloc < 1:
            raise ValueError("sharded FastEGNN needs at least one node per rank")
        self.n_src = world * self.Npad      # global ids index the padded, all-gathered table directly

    def rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.n0:self.n1].contiguous()

    def edges(self, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor]):
        m = (edge_index[0] >= self.n0) & (edge_index[0] < self.n1)
        ei = edge_index[:, m].contiguous()
        return ei, (edge_attr[m].contiguous() if edge_attr is not None else None)


def _layer_lists(spec: _Spec, params, i):
    return [params[s] if s is not None else None for s in spec.layer_slots[i]]


class _ShardedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, group, spec: _Spec, plan: ShardPlan, graph, batch32, gptr, ea_sorted, node_attr,
                node_feat, node_loc, node_vel, loc_mean, *params):
        W, rank = plan.world, plan.rank
        N, Npad, B, Cn = plan.nloc, plan.Npad, loc_mean.size(0), spec.C
        params = [p.detach() for p in params]
        node_feat, node_loc, node_vel, loc_mean = (t.detach().contiguous().float()
                                                   for t in (node_feat, node_loc, node_vel, loc_mean))
        h = be.empty(N, H)
        be.embed_forward(node_feat, spec.nf, params[1], params[2], h)
        HvT = be.empty(B, Cn, H)
        be.virtual_init(params[0], B, Cn, HvT)
        x, Z = node_loc, loc_mean
        saved = []
        for i in range(spec.n_layers):
            lp = _layer_lists(spec, params, i)
            b = dict(h=h, x=x, Z=Z, HvT=HvT)
            b.update(be.carve(dict(wpack=(be.wpack_floats(Cn),), P=(N, H), A=(N, H), svel=(N,), sgrav=(N,), xsum=(B, 4),
                                   Bc=(B, Cn, H), aggm=(N, H), npre=(N, H), aggx=(N, 3))))
            b["QX"] = be.zeros(Npad, K.QX_LD)                      # padded: equal-sized all-gather shards
            b["QX_src"] = be.empty(W * Npad, K.QX_LD)
            nV = B * Cn * H
            pools = be.empty(nV + B * 3 * Cn)                      # poolV | poolX adjacent: one all-reduce
            b["poolV"], b["poolX"] = pools[:nV].view(B, Cn, H), pools[nV:].view(B, 3, Cn)
            b.update(h_out=be.empty(N, H), x_out=be.empty(N, 3), Z_out=be.empty(B, 3, Cn), HvT_out=be.empty(B, Cn, H))
            t = dict(batch=batch32, gptr=gptr, vel=node_vel, ea_sorted=ea_sorted, node_attr=node_attr, **b)
            be.stage("pack_weights", spec, N, B, graph, t, lp)
            be.stage("node_pre_forward", spec, N, B, graph, t, lp)
            dist.all_gather_into_tensor(b["QX_src"], b["QX"], group=group)
            be.stage("graph_xsum", spec, N, B, graph, t, lp)
            dist.all_reduce(b["xsum"], group=group)
            be.stage("graph_pre_forward", spec, N, B, graph, t, lp)
            be.stage("edge_forward", spec, N, B, graph, t, lp)
            be.stage("virt_forward", spec, N, B, graph, t, lp)
            dist.all_reduce(pools, group=group)
            be.stage("graph_post_forward", spec, N, B, graph, t, lp)
            saved.append(b)
            h, x, Z, HvT = b["h_out"], b["x_out"], b["Z_out"], b["HvT_out"]
            for k in ("aggx", "poolX", "h_out", "x_out", "Z_out", "HvT_out"):
                del b[k]
        ctx.be, ctx.group, ctx.spec, ctx.plan, ctx.graph, ctx.saved = be, group, spec, plan, graph, saved
        ctx.misc = (batch32, gptr, ea_sorted, node_attr, node_feat, node_vel, params)
        return x, Z

    @staticmethod
    def backward(ctx, g_loc, g_vloc):
        be, group, spec, plan, graph, saved = ctx.be, ctx.group, ctx.spec, ctx.plan, ctx.graph, ctx.saved
        batch32, gptr, ea_sorted, node_attr, node_feat, node_vel, params = ctx.misc
        W, rank = plan.world, plan.rank
        N, Npad, B, Cn, E = plan.nloc, plan.Npad, saved[0]["Z"].size(0), spec.C, graph.E
        grads = [be.zeros(*p.shape) for p in params]
        # the per-graph stages run on every rank; only rank 0 keeps their weight gradients
        dummy = grads if rank == 0 else [be.zeros(*p.shape) for p in params]
        g_h = be.zeros(N, H)
        g_x = (g_loc if g_loc is not None else be.zeros(N, 3)).contiguous().float()
        g_Z = (g_vloc if g_vloc is not None else be.zeros(B, 3, Cn)).contiguous().float()
        g_HvT = be.zeros(B, Cn, H)
        g_vel = be.zeros(N, 3)
        M = max(N, B * Cn)
        sc = be.carve(dict(g_poolV=(B, Cn, H), g_poolX=(B, 3, Cn), g_xbar=(B, 4), g_A=(N, H), g_P=(N, H),
                           g_aggm=(N, H), g_aggx=(N, 3), g_svel=(N,), g_sgrav=(N,), g_QXe=(max(E, 1), K.QX_LD),
                           g_xrow=(N, 3), wg_edge=(be.wg_edge_floats(E),), wg_virt=(5 * N * Cn * H,),
                           wg_node=(8 * M * H,), wg_slab=(be.wg_slab_floats(),)))
        nV = B * Cn * H
        gpools = be.empty(nV + B * 3 * Cn)                                 # g_Bc | g_Zp adjacent: one all-reduce
        sc["g_Bc"], sc["g_Zp"] = gpools[:nV].view(B, Cn, H), gpools[nV:].view(B, 3, Cn)
        sc["g_QX_src"] = be.empty(W * Npad, K.QX_LD)
        sc["g_QX"] = be.empty(Npad, K.QX_LD)
        for i in reversed(range(spec.n_layers)):
            b = saved[i]
            lp = _layer_lists(spec, params, i)
            lg = _layer_lists(spec, grads, i)
            ld = _layer_lists(spec, dummy, i)
            out = dict(g_h=be.empty(N, H), g_x=be.empty(N, 3), g_Z=be.empty(B, 3, Cn), g_HvT=be.empty(B, Cn, H))
            t = dict(batch=batch32, gptr=gptr, vel=node_vel, ea_sorted=ea_sorted, node_attr=node_attr,
                     g_h_out=g_h, g_x_out=g_x, g_Z_out=g_Z, g_HvT_out=g_HvT, g_vel=g_vel, **b, **out, **sc)
            be.stage("graph_post_backward", spec, N, B, graph, t, lp, ld)
            be.stage("virt_backward", spec, N, B, graph, t, lp, lg)
            dist.all_reduce(gpools, group=group)
            be.stage("graph_pre_backward", spec, N, B, graph, t, lp, ld)
            be.stage("edge_backward", spec, N, B, graph, t, lp, lg)
            be.stage("edge_col_reduce", spec, N, B, graph, t, lp, lg)
            dist.reduce_scatter_tensor(sc["g_QX"], sc["g_QX_src"], group=group)
            be.stage("node_pre_backward", spec, N, B, graph, t, lp, lg)
            g_h, g_x, g_Z, g_HvT = out["g_h"], out["g_x"], out["g_Z"], out["g_HvT"]
            saved[i] = None
        be.virtual_init_backward(g_HvT, B, Cn, dummy[0])
        g_nf = torch.empty_like(node_feat) if ctx.needs_input_grad[9] else None
        be.embed_backward(node_feat, g_h, spec.nf, params[1], grads[1], grads[2], g_nf)
        return (None,) * 9 + (g_nf, g_x, g_vel, g_Z, *grads)


class ShardedFastEGNN(torch.nn.Module):
    """Wraps a (replicated) FastEGNN so that ONE batch is evaluated cooperatively by all ranks.

    forward(...) takes the full-graph inputs (every rank passes the same tensors, like a replicated data
    loader would) and returns (node_loc rows owned by this rank, virtual_node_loc replicated).  After
    loss.backward() call ``fastegnn_amd.dist.allreduce_gradients(model.parameters())``.
    """

    def __init__(self, model: FastEGNN, group=None, backend=None):
        super().__init__()
        self.model = model
        self.group = group
        self.backend = backend

    def forward(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None,
                node_attr=None):
        m = self.model
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        be = self.backend or HipBackend(node_loc.device)
        if m._spec is None:
            m._spec = _Spec(m)
            pidx = m._param_index
            m._plist = [pidx[n] for n in m._spec.names]
        spec = m._spec
        plan = ShardPlan(node_loc.size(0), world, rank)
        ei, ea = plan.edges(edge_index, edge_attr.detach() if edge_attr is not None else None)
        graph = be.build_graph(ei, plan.nloc, plan.n_src, plan.n0)
        ea_sorted = graph.permute(ea)
        B = loc_mean.size(0)
        batch32, gptr = be.build_batch(plan.rows(data_batch), plan.nloc, B)
        na = plan.rows(node_attr.detach()).float() if node_attr is not None else None
        self.plan = plan
        return _ShardedFunction.apply(be, self.group, spec, plan, graph, batch32, gptr, ea_sorted, na,
                                      plan.rows(node_feat), plan.rows(node_loc), plan.rows(node_vel), loc_mean,
                                      *m._plist)
