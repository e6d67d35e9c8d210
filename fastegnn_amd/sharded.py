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

    def wg_virt_floats(self, N, Cn):
        return self.lib.fastegnn_wg_virt_floats(N, Cn)

    def wg_node_floats(self, N, B, Cn):
        return self.lib.fastegnn_wg_node_floats(N, B, Cn)

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


class CommStats:
    """Per-collective byte and time totals of the sharded path (bench.py): bytes are the payload every rank
    contributes (all-gather / reduce-scatter: the full gathered table), time is issue-to-completion on this rank,
    measured with events on the communicator's side of the async work handle (so overlap with kernels that were
    launched behind the collective is NOT subtracted)."""

    def __init__(self):
        self.calls: Dict[str, int] = {}
        self.bytes: Dict[str, int] = {}
        self.events: Dict[str, list] = {}

    def summary(self, steps: int = 1):
        out = {}
        for k in self.calls:
            ms = 0.0
            for a, b in self.events.get(k, []):
                b.synchronize()
                ms += a.elapsed_time(b)
            out[k] = {"calls_per_step": self.calls[k] / steps, "bytes_per_step": self.bytes[k] / steps,
                      "ms_per_step": round(ms / steps, 4)}
        return out


class _Done:
    def wait(self):
        return True


class _Timed:
    """Async work handle that records a completion event on the current stream right after the wait."""

    def __init__(self, work, rec):
        self.work, self.rec = work, rec

    def wait(self):
        self.work.wait()
        if self.rec is not None:
            self.rec[1].record()
        return True


class _Comm:
    """The three exchange steps of SURVEY 8e as asynchronous collectives (torch.distributed: RCCL on the GPU,
    gloo in the CPU tests).  World size 1 short-circuits to local copies."""

    def __init__(self, group, stats: Optional[CommStats]):
        self.group, self.stats = group, stats
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a transport without device support (gloo): device tensors are staged through host memory, synchronously
        self.host_staged = self.world > 1 and dist.get_backend(group) == "gloo"

    def _staged(self, fn, out, *inputs):
        h_in = [t.cpu() for t in inputs]
        h_out = out.cpu() if not inputs else torch.empty(out.shape, dtype=out.dtype)
        fn(h_out, *h_in)
        out.copy_(h_out)
        return _Done()

    def _note(self, name, t):
        rec = None
        if self.stats is not None:
            st = self.stats
            st.calls[name] = st.calls.get(name, 0) + 1
            st.bytes[name] = st.bytes.get(name, 0) + t.numel() * t.element_size()
            if t.is_cuda:
                rec = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                rec[0].record()
                st.events.setdefault(name, []).append(rec)
        return rec

    def all_reduce(self, name, t):
        rec = self._note(name, t)
        if self.world == 1:
            return _Timed(_Done(), rec)
        if self.host_staged and t.is_cuda:
            return _Timed(self._staged(lambda o: dist.all_reduce(o, group=self.group), t), rec)
        return _Timed(dist.all_reduce(t, group=self.group, async_op=True), rec)

    def all_gather(self, name, out, inp):
        rec = self._note(name, out)
        if self.world == 1:
            out.copy_(inp)
            return _Timed(_Done(), rec)
        if self.host_staged and out.is_cuda:
            return _Timed(self._staged(lambda o, i: dist.all_gather_into_tensor(o, i, group=self.group), out, inp), rec)
        return _Timed(dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True), rec)

    def reduce_scatter(self, name, out, inp):
        rec = self._note(name, inp)
        if self.world == 1:
            out.copy_(inp)
            return _Timed(_Done(), rec)
        if self.host_staged and out.is_cuda:
            return _Timed(self._staged(lambda o, i: dist.reduce_scatter_tensor(o, i, group=self.group), out, inp), rec)
        return _Timed(dist.reduce_scatter_tensor(out, inp, group=self.group, async_op=True), rec)


def _layer_lists(spec: _Spec, params, i):
    return [params[s] if s is not None else None for s in spec.layer_slots[i]]


class _ShardedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, group, stats, spec: _Spec, plan: ShardPlan, graph, batch32, gptr, ea_sorted, node_attr,
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
        comm = _Comm(group, stats)
        # Collectives are issued asynchronously (RCCL runs them on its own stream) and waited for at their first
        # consumer, so that every exchange has independent kernels in flight behind it:
        #   all-reduce xsum_i      behind  node_pre_i
        #   all-gather QX_i        behind  graph_post_{i-1} (deferred) and graph_pre_i
        #   all-reduce pools_i     behind  pack_{i+1}, graph_xsum_{i+1}, node_pre_{i+1}
        # (all ranks issue them in the same order: pools_{i-1}, xsum_i, QX_i)
        pend = None     # (work, t, lp, b) of the previous layer: its graph_post is still to run
        for i in range(spec.n_layers):
            lp = _layer_lists(spec, params, i)
            b = dict(h=h, x=x)
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
            be.stage("graph_xsum", spec, N, B, graph, t, lp)
            w_xsum = comm.all_reduce("xsum", b["xsum"])
            be.stage("node_pre_forward", spec, N, B, graph, t, lp)
            w_qx = comm.all_gather("QX", b["QX_src"], b["QX"])
            if pend is not None:                                   # virtual state of this layer <- previous layer's pools
                pend[0].wait()
                be.stage("graph_post_forward", spec, N, B, graph, pend[1], pend[2])
                Z, HvT = pend[3]["Z_out"], pend[3]["HvT_out"]
                for k in ("aggx", "poolX", "h_out", "x_out", "Z_out", "HvT_out"):
                    del pend[3][k]
            b["Z"], b["HvT"] = Z, HvT
            t["Z"], t["HvT"] = Z, HvT
            w_xsum.wait()
            be.stage("graph_pre_forward", spec, N, B, graph, t, lp)
            w_qx.wait()
            be.stage("edge_forward", spec, N, B, graph, t, lp)
            be.stage("virt_forward", spec, N, B, graph, t, lp)
            pend = (comm.all_reduce("pools", pools), t, lp, b)
            saved.append(b)
            h, x = b["h_out"], b["x_out"]
        pend[0].wait()
        be.stage("graph_post_forward", spec, N, B, graph, pend[1], pend[2])
        Z = pend[3]["Z_out"]
        for k in ("aggx", "poolX", "h_out", "x_out", "Z_out", "HvT_out"):
            del pend[3][k]
        ctx.be, ctx.group, ctx.spec, ctx.plan, ctx.graph, ctx.saved = be, group, spec, plan, graph, saved
        ctx.comm = comm
        ctx.misc = (batch32, gptr, ea_sorted, node_attr, node_feat, node_vel, params)
        return x, Z

    @staticmethod
    def backward(ctx, g_loc, g_vloc):
        be, group, spec, plan, graph, saved = ctx.be, ctx.group, ctx.spec, ctx.plan, ctx.graph, ctx.saved
        comm = ctx.comm
        batch32, gptr, ea_sorted, node_attr, node_feat, node_vel, params = ctx.misc
        W, rank = plan.world, plan.rank
        N, Npad, B, Cn, E = plan.nloc, plan.Npad, saved[0]["Z"].size(0), spec.C, graph.E
        # gradient buffers: ONE zero-filled allocation carved into 16-byte aligned views (one fill launch instead of one
        # per parameter; the slices become the .grad tensors, so fastegnn_amd.dist.allreduce_gradients reduces the flat
        # buffer in place)
        def flat_like(ps):
            sizes = [(p.numel() + 3) // 4 * 4 for p in ps]
            flat = be.zeros(sum(sizes))
            out, off = [], 0
            for p, n in zip(ps, sizes):
                out.append(flat[off:off + p.numel()].view(p.shape))
                off += n
            return out
        grads = flat_like(params)
        # the per-graph stages run on every rank; only rank 0 keeps their weight gradients
        dummy = grads if rank == 0 else flat_like(params)
        g_h = be.zeros(N, H)
        g_x = (g_loc if g_loc is not None else be.zeros(N, 3)).contiguous().float()
        g_Z = (g_vloc if g_vloc is not None else be.zeros(B, 3, Cn)).contiguous().float()
        g_HvT = be.zeros(B, Cn, H)
        g_vel = be.zeros(N, 3)
        sc = be.carve(dict(g_poolV=(B, Cn, H), g_poolX=(B, 3, Cn), g_xbar=(B, 4), g_A=(N, H), g_P=(N, H),
                           g_aggm=(N, H), g_aggx=(N, 3), g_svel=(N,), g_sgrav=(N,), g_QXe=(max(E, 1), K.QX_LD),
                           g_xrow=(N, 3), wg_edge=(be.wg_edge_floats(E),), wg_virt=(be.wg_virt_floats(N, Cn),),
                           wg_node=(be.wg_node_floats(N, B, Cn),), wg_slab=(be.wg_slab_floats(),)))
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
            w_pools = comm.all_reduce("g_pools", gpools)           # behind the edge backward
            be.stage("edge_backward", spec, N, B, graph, t, lp, lg)
            be.stage("edge_col_reduce", spec, N, B, graph, t, lp, lg)
            w_qx = comm.reduce_scatter("g_QX", sc["g_QX"], sc["g_QX_src"])   # behind graph_pre_backward
            w_pools.wait()
            be.stage("graph_pre_backward", spec, N, B, graph, t, lp, ld)
            w_qx.wait()
            be.stage("node_pre_backward", spec, N, B, graph, t, lp, lg)
            g_h, g_x, g_Z, g_HvT = out["g_h"], out["g_x"], out["g_Z"], out["g_HvT"]
            saved[i] = None
        be.virtual_init_backward(g_HvT, B, Cn, dummy[0])
        g_nf = torch.empty_like(node_feat) if ctx.needs_input_grad[10] else None
        be.embed_backward(node_feat, g_h, spec.nf, params[1], grads[1], grads[2], g_nf)
        return (None,) * 10 + (g_nf, g_x, g_vel, g_Z, *grads)


class ShardedFastEGNN(torch.nn.Module):
    """Wraps a (replicated) FastEGNN so that ONE batch is evaluated cooperatively by all ranks.

    ``forward(...)`` takes the full-graph inputs (every rank passes the same tensors, like a replicated data
    loader would) and returns (node_loc rows owned by this rank, virtual_node_loc replicated).  A caller that
    shards at load time passes the result of ``shard_inputs(...)`` to ``forward_local`` instead, so that no rank
    ever holds (or filters) the full COO.

    Loss contract: a loss term on ``node_loc`` covers the LOCAL rows only (``plan.rows(target)``; the ranks'
    terms add up to the global loss); a loss term on ``virtual_node_loc`` must be the FULL term, identical on
    every rank -- the virtual state is replicated, the per-graph stages run redundantly and only rank 0 keeps
    their weight gradients, so the gradient arriving at ``virtual_node_loc`` is not summed over ranks.  After
    ``loss.backward()`` call ``fastegnn_amd.dist.allreduce_gradients(model.parameters())``.
    """

    def __init__(self, model: FastEGNN, group=None, backend=None, stats: Optional[CommStats] = None):
        super().__init__()
        if model.hidden_nf != K.H:   # the zero-padded path of model.py (_pad_param) is single-GPU only
            raise NotImplementedError(f"fastegnn_amd.ShardedFastEGNN: hidden_nf must be {K.H}")
        self.model = model
        self.group = group
        self.backend = backend
        self.stats = stats
        self.plan: Optional[ShardPlan] = None

    def _world_rank(self):
        if not dist.is_initialized():
            return 1, 0
        return dist.get_world_size(self.group), dist.get_rank(self.group)

    def shard_inputs(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None,
                     node_attr=None) -> Dict[str, torch.Tensor]:
        """This rank's share of a batch: its node rows, the edges aggregating into them (global column ids),
        the replicated per-graph tensors."""
        for name, t in (("edge_attr", edge_attr), ("node_attr", node_attr)):
            if t is not None and t.requires_grad:   # the single-GPU module returns these; the sharded caller does not (yet)
                raise NotImplementedError(f"fastegnn_amd.ShardedFastEGNN: gradient w.r.t. {name} is not implemented")
        world, rank = self._world_rank()
        plan = ShardPlan(node_loc.size(0), world, rank)
        ei, ea = plan.edges(edge_index, edge_attr.detach() if edge_attr is not None else None)
        return dict(plan=plan, node_feat=plan.rows(node_feat), node_loc=plan.rows(node_loc), node_vel=plan.rows(node_vel),
                    edge_index=ei, edge_attr=ea, data_batch=plan.rows(data_batch), loc_mean=loc_mean,
                    node_attr=plan.rows(node_attr.detach()).float() if node_attr is not None else None)

    def forward_local(self, local: Dict[str, torch.Tensor]):
        m = self.model
        plan: ShardPlan = local["plan"]
        be = self.backend or HipBackend(local["node_loc"].device)
        if m._spec is None:
            m._spec = _Spec(m)
            pidx = m._param_index
            m._plist = [pidx[n] for n in m._spec.names]
        spec = m._spec
        graph = be.build_graph(local["edge_index"], plan.nloc, plan.n_src, plan.n0)
        ea_sorted = graph.permute(local["edge_attr"])
        B = local["loc_mean"].size(0)
        batch32, gptr = be.build_batch(local["data_batch"], plan.nloc, B)
        self.plan = plan
        return _ShardedFunction.apply(be, self.group, self.stats, spec, plan, graph, batch32, gptr, ea_sorted,
                                      local["node_attr"], local["node_feat"], local["node_loc"], local["node_vel"],
                                      local["loc_mean"], *m._plist)

    def forward(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None,
                node_attr=None):
        return self.forward_local(self.shard_inputs(node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean,
                                                    edge_attr, node_attr))
