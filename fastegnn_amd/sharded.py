"""One large graph sharded over the ranks of a node ("row-owner" scheme).

Rank r owns the contiguous node range [r*Npad, (r+1)*Npad) and every edge whose aggregation row lies
in it, so all segment means are complete locally.  Per layer the ranks exchange exactly:

  forward   table exchange QX [nloc,68] -> QX_src              (the rows of the source table this rank's edges read)
              "halo" (default): all-to-all-v of the GHOST rows only -- the remote nodes this rank's edges point at,
                      found once per graph (HaloPlan); QX_src = [own rows | ghost rows], column ids remapped.  With
                      spatially ordered nodes (shard_inputs(reorder=True): Morton order inside each graph) a rank's
                      ghosts are the shell of its region: 2-6 % of the all-gather's bytes at cfg4 / cfg5 on 8 ranks
              "allgather": all-gather of the padded shards, QX_src [W*Npad,68] indexed by global id
            all-reduce   xsum [B,4]                            (centroids / node counts)
            all-reduce   poolV|poolX [B,C,64]+[B,3,C]          (virtual-node accumulators)
  backward  all-reduce   g_Bc|g_Zp                             (adjoint of the broadcast virtual state)
            transpose of the table exchange: ghost-row gradients back to their owners (all-to-all-v + scatter-add),
              or reduce-scatter g_QX_src [W*Npad,68] -> g_QX [Npad,68]
  once      all-reduce   parameter gradients (fastegnn_amd.dist.allreduce_gradients, by the caller)

Schedule (round 4).  A rank's rows are ordered [interior | boundary]: a BOUNDARY row has at least one edge from a ghost
column, an interior row reads own rows only (HaloPlan.build finds the split and renumbers the rank's rows; a rank whose
rows lie in several graphs keeps its order and has no interior part).  The edge stage runs as two launches over the two
row ranges: forward  node_pre -> [halo exchange] || graph_pre, edge(interior) -> edge(boundary); backward
edge(boundary) -> [ghost gradients back] || edge(interior), graph_pre -> node_pre.  The centroid sums of layer l+1 travel
in the same all-reduce as the pools of layer l (x_out of layer l is final after its virt stage), so a layer costs ONE
small all-reduce per direction; the backward stages of a layer queue their weight-gradient contractions into one batch
(fastegnn_wgrad_batch_*: one contraction launch + one reduction launch per layer, as the unsharded layer call has).
With the C-ABI transport (FASTEGNN_COMM=abi) and FASTEGNN_SHARDED_SYNC=0 the collectives run on a second HIP stream,
forked and joined with events -- a pattern a HIP graph capture records as parallel branches -- otherwise on the compute
stream in program order (the default until RCCL with more than one rank has run on real hardware).

The virtual state (Z, Hv) is replicated and updated identically everywhere; the per-graph stages
(graph_pre / graph_post) run redundantly on every rank and only rank 0 contracts their weight gradients.
The compute of every stage is one C-ABI call (include/fastegnn_hip.h); this file is only the
orchestration, written against a small backend interface so that the world_size-2 gloo test can drive
it on CPU with the oracle's stage functions (tests/test_sharded_cpu.py).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _lib as K
from .model import FastEGNN, SortedGraph, _PadParams, _PtrTable, _Spec, _carve, _fill, _new_layer, _stream

H = K.H


# ------------------------------------------------------------------------------------------
# backend: everything that touches device memory / kernels
# ------------------------------------------------------------------------------------------
class HipBackend:
    """Stage calls on libfastegnn_hip.so (the product path)."""

    def __init__(self, device, act: bool = False, wide: bool = False):
        self.dev = device
        self.lib = K.lib(act, wide=wide)   # act: the generic-activation build (a model whose act_fn is not SiLU); wide: its bf16x3 form

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

    def wg_virt_floats(self, N, Cn, flags=0):
        return self.lib.fastegnn_wg_virt_floats_for(N, Cn, flags)

    def wg_node_floats(self, N, B, Cn):
        return self.lib.fastegnn_wg_node_floats(N, B, Cn)

    def build_graph(self, edge_index, n_rows, n_src, row_begin, csc=True):
        return SortedGraph(edge_index, n_rows, n_src, row_begin, csc=csc)

    def pad_params(self, names, h, C_, rf, params):
        """hidden_nf < 64: the parameters' 64-wide images (fastegnn_pad_params, differentiable)"""
        return list(_PadParams.apply(tuple(names), h, C_, rf, *params))

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

    def pack_all(self, spec, N, B, graph, layer_params, wpacks):
        """the weight images of every layer in one launch (fastegnn_pack_weights_all)"""
        n = len(layer_params)
        arr = (C.POINTER(K.LayerT) * n)()
        keep = []
        for i in range(n):
            L = _new_layer(spec, N, B, graph)
            ptab = _PtrTable(layer_params[i])
            _fill(L, params=ptab.addr(), wpack=wpacks[i])
            keep.append((L, ptab))
            arr[i] = C.pointer(L)
        K.check(self.lib.fastegnn_pack_weights_all(arr, n, _stream(self.dev)), "fastegnn_pack_weights_all")

    def wgrad_open(self, spec, N, B, graph, t, params):
        """One weight-gradient batch for the backward stages of a layer (fastegnn_wgrad_batch_open)."""
        L = _new_layer(spec, N, B, graph)
        ptab = _PtrTable(params)
        _fill(L, params=ptab.addr(), wg_slab=t["wg_slab"])
        h = C.c_void_p()
        K.check(self.lib.fastegnn_wgrad_batch_open(C.byref(L), _stream(self.dev), C.byref(h)), "fastegnn_wgrad_batch_open")
        return h.value

    def wgrad_close(self, handle):
        if handle:
            K.check(self.lib.fastegnn_wgrad_batch_close(C.c_void_p(handle)), "fastegnn_wgrad_batch_close")

    def stage(self, name, spec, N, B, graph, t: Dict[str, torch.Tensor], params, grads=None, flags=0):
        L = _new_layer(spec, N, B, graph)
        L.flags |= flags
        ptab = params.ptab() if isinstance(params, _LayerList) else _PtrTable(params)
        keep = [ptab]
        _fill(L, params=ptab.addr(), **{k: v for k, v in t.items() if v is not None})
        if grads is not None:
            gtab = grads.ptab() if isinstance(grads, _LayerList) else _PtrTable(grads)
            keep.append(gtab)
            _fill(L, grads=gtab.addr())
        K.check(getattr(self.lib, "fastegnn_" + name)(C.byref(L), _stream(self.dev)), "fastegnn_" + name)


# ------------------------------------------------------------------------------------------
# partition
# ------------------------------------------------------------------------------------------
class ShardPlan:
    """Contiguous node ranges of equal padded size Npad = ceil(N / W); table exchange by all-gather."""

    mode = "allgather"

    def __init__(self, n_nodes: int, world: int, rank: int, order: Optional[torch.Tensor] = None):
        self.N, self.world, self.rank = n_nodes, world, rank
        self.Npad = (n_nodes + world - 1) // world
        self.n0 = min(n_nodes, rank * self.Npad)
        self.n1 = min(n_nodes, self.n0 + self.Npad)
        self.nloc = self.n1 - self.n0
        if self.nloc < 1:
            raise ValueError("sharded FastEGNN needs at least one node per rank")
        self.n_src = world * self.Npad      # global ids index the padded, all-gathered table directly
        self.n_table = self.Npad            # rows of the local table buffer (padded: equal-sized all-gather shards)
        # order[new] = caller's node id (None: identity): the plan's ranges are ranges of the REORDERED nodes
        self.order = order
        self.node_ids = order[self.n0:self.n1] if order is not None else None
        # row ranges the edge stage is launched over: (first local row, rows, needs the exchanged table)
        self.parts = [(0, self.nloc, True)]
        self.edge_perm = None      # HaloPlan.build: order of the rank's edges by part (None: as given)
        self.edge_counts = None    # edges per part, in that order

    def rows(self, t: torch.Tensor) -> torch.Tensor:
        """This rank's rows of a per-node tensor given in the CALLER's node order."""
        if self.node_ids is not None:
            return t.index_select(0, self.node_ids.to(t.device)).contiguous()
        return t[self.n0:self.n1].contiguous()

    def edges(self, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor]):
        """Edges (in the plan's node numbering) whose aggregation row this rank owns."""
        m = (edge_index[0] >= self.n0) & (edge_index[0] < self.n1)
        ei = edge_index[:, m].contiguous()
        return ei, (edge_attr[m].contiguous() if edge_attr is not None else None)

    # -- table exchange (forward) and its transpose (backward); `t`: the layer's buffers
    def alloc_tables(self, be):
        QX = be.zeros(self.Npad, K.QX_LD)                      # padded: equal-sized all-gather shards
        return QX, be.empty(self.world * self.Npad, K.QX_LD)

    def exchange_forward(self, comm, QX, QX_src):
        return comm.all_gather("QX", QX_src, QX)

    def alloc_grad_tables(self, be):
        return be.empty(self.world * self.Npad, K.QX_LD), be.empty(self.Npad, K.QX_LD)

    def exchange_backward(self, comm, g_QX_src, g_QX, deterministic: bool = False):
        return comm.reduce_scatter("g_QX", g_QX, g_QX_src)

    def exchanged_bytes(self):
        """Bytes this rank receives per table exchange (one layer, one direction)."""
        return (self.world - 1) * self.Npad * K.QX_LD * 4


class _HaloWork:
    """Completion of a halo exchange: wait for the all-to-all-v, then (backward) add the returned rows to their owners."""

    def __init__(self, work, after=None):
        self.work, self.after = work, after

    def wait(self):
        self.work.wait()
        if self.after is not None:
            self.after()
        return True


class HaloPlan(ShardPlan):
    """Same contiguous ownership, but only the GHOST rows travel: the remote nodes this rank's edges point at.

    Built once per graph by `build()` (collective: every rank of the group must call it): the ranks tell each other which
    of their rows they need (one all-to-all of counts, one all-to-all-v of ids).  Source table of this rank:
    [own rows (nloc) | ghost rows, grouped by owner rank in ascending id]; the edge columns are remapped to it.
    Forward: pack the rows each peer asked for, all-to-all-v, the ghosts land behind the own rows.  Backward: the ghost
    part of the col-keyed gradient goes back the same way and is added to the owners' rows."""

    mode = "halo"

    def build(self, edge_index: torch.Tensor, comm, local_batch: Optional[torch.Tensor] = None, split: bool = True,
              peers_want: Optional[torch.Tensor] = None) -> torch.Tensor:
        """edge_index: this rank's edges (rows in [n0,n1), cols = global ids in the plan's numbering).  Returns the edge
        index with the rows renumbered [interior | boundary] (still offset by n0) and the columns remapped to the local
        source table.  local_batch: graph id of the rank's rows (the split needs them to lie in ONE graph: data_batch must
        stay ascending); split=False keeps the row order (one launch that waits for the halo).  peers_want: emulation of
        one rank of a larger world in a single process (bench.py --emulate-world): the global ids of this rank's rows its
        peers would ask for, grouped by peer, instead of the id exchange."""
        W, dev = self.world, edge_index.device
        rows, cols = edge_index[0], edge_index[1]
        remote = (cols < self.n0) | (cols >= self.n1)
        ghost = torch.unique(cols[remote])                                 # ascending => grouped by owner (contiguous ranges)
        owner = torch.div(ghost, self.Npad, rounding_mode="floor")
        recv_counts = torch.bincount(owner, minlength=W).to(torch.int64)   # rows this rank receives from each owner
        if peers_want is None:
            send_counts = comm.exchange_counts(recv_counts)                # rows each peer wants from this rank
            self.recv_splits = [int(v) for v in recv_counts.tolist()]
            self.send_splits = [int(v) for v in send_counts.tolist()]
            want = comm.exchange_ids(ghost, self.send_splits, self.recv_splits)   # global ids of MY rows, grouped by asking peer
        else:
            want = peers_want
            self.recv_splits = [int(v) for v in recv_counts.tolist()]
            self.send_splits = None
        if want.numel() and (int(want.min()) < self.n0 or int(want.max()) >= self.n1):
            raise RuntimeError("HaloPlan: a peer asked for a row this rank does not own")
        # [interior | boundary] order of the own rows (stable: the Morton locality inside each part survives)
        is_b = torch.zeros(self.nloc, dtype=torch.bool, device=dev)
        is_b[rows[remote] - self.n0] = True
        one_graph = local_batch is None or local_batch.numel() == 0 or bool(local_batch[0] == local_batch[-1])
        if split and one_graph and bool(is_b.any()):
            lperm = torch.argsort(is_b.to(torch.int8), stable=True)        # lperm[new] = old local row
            n_int = int((~is_b).sum())
            self.parts = [p for p in ((0, n_int, False), (n_int, self.nloc - n_int, True)) if p[1] > 0]
        else:
            lperm = torch.arange(self.nloc, device=dev)
            self.parts = [(0, self.nloc, bool(is_b.any()))]
        self.n_int = sum(n for _, n, halo in self.parts if not halo)
        lrank = torch.empty_like(lperm)
        lrank[lperm] = torch.arange(self.nloc, device=dev)                 # lrank[old] = new
        base = self.order[self.n0:self.n1].to(dev) if self.order is not None else torch.arange(self.n0, self.n1, device=dev)
        self.node_ids = base[lperm]                                        # caller's node id of every local row
        self.send_ids = lrank[want - self.n0].contiguous()
        self.n_ghost = int(ghost.numel())
        self.n_send = int(self.send_ids.numel())
        self.n_src = self.nloc + self.n_ghost
        self.n_table = self.nloc
        pos = torch.searchsorted(ghost, cols.clamp(min=0)) if self.n_ghost else torch.zeros_like(cols)
        own = lrank[(cols - self.n0).clamp(0, self.nloc - 1)]
        new_col = torch.where(remote, self.nloc + pos, own)
        new_row = lrank[rows - self.n0]
        ei = torch.stack([new_row + self.n0, new_col])
        if len(self.parts) > 1:
            # the rank's edges grouped by part (stable), so that a part's edges -- and its rows of edge_attr -- are a contiguous
            # slice: no data-dependent selection inside the step (a boolean mask is a host synchronisation, which a HIP graph
            # capture of the step does not allow)
            in_b = new_row >= self.n_int
            self.edge_perm = torch.argsort(in_b.to(torch.int8), stable=True)
            n_b = int(in_b.sum())
            self.edge_counts = [ei.size(1) - n_b, n_b]
            ei = ei[:, self.edge_perm]
        else:
            self.edge_perm, self.edge_counts = None, [ei.size(1)]
        return ei.contiguous()

    def alloc_tables(self, be):
        QX_src = be.zeros(max(self.n_src, 1), K.QX_LD)
        return QX_src[:self.nloc], QX_src          # node_pre writes the own rows in place: no copy

    def exchange_forward(self, comm, QX, QX_src):
        if self.world == 1:             # (never skipped on a rank-local condition: the all-to-all-v is a collective)
            return _Done()
        if self.send_splits is None:     # emulated rank of a larger world: pack as a real rank would, fill the ghost rows locally
            return comm.emulated_exchange("QX_halo", QX_src[self.nloc:], QX, self.send_ids)
        return comm.halo_exchange("QX_halo", QX_src[self.nloc:], QX, self.send_ids, self.recv_splits, self.send_splits)

    def alloc_grad_tables(self, be):
        g_src = be.empty(max(self.n_src, 1), K.QX_LD)
        return g_src, g_src[:self.nloc]            # node_pre_bwd reads the own rows in place

    def exchange_backward(self, comm, g_QX_src, g_QX, deterministic: bool = False):
        if self.world == 1:
            return _Done()
        back = g_QX_src.new_empty(self.n_send, K.QX_LD)
        send = g_QX_src[self.nloc:self.nloc + self.n_ghost]
        if self.send_splits is None:
            work = comm.emulated_return("g_QX_halo", back, send)
        else:
            work = comm.all_to_all_v("g_QX_halo", back, send, self.send_splits, self.recv_splits)

        def add():
            if not self.n_send:
                return
            if deterministic and self.send_splits is not None:
                # FASTEGNN_F_DETERMINISTIC: a row may have been asked for by several peers; peer by peer in rank order, and
                # inside a peer's block every id occurs once (no colliding atomics): the sum has a fixed order
                off = 0
                for n in self.send_splits:
                    if n:
                        ids = self.send_ids[off:off + n]
                        g_QX[ids] = g_QX[ids] + back[off:off + n]
                    off += n
            else:
                comm.scatter_add_rows(g_QX, self.send_ids, back)
        return _HaloWork(work, add)

    def exchanged_bytes(self):
        return self.n_ghost * K.QX_LD * 4


def morton_order(loc: torch.Tensor, data_batch: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """order[new] = old: nodes sorted by (graph, Morton code of their position on a 2^bits grid over the bounding box).
    data_batch stays ascending; contiguous index ranges become compact regions of space, so a rank's ghosts are the
    shell of its region instead of almost every remote node."""
    lo = loc.min(0).values
    span = (loc.max(0).values - lo).clamp(min=1e-30)
    g = ((loc - lo) / span * (2 ** bits - 1)).long().clamp(0, 2 ** bits - 1)
    code = torch.zeros(loc.size(0), dtype=torch.int64, device=loc.device)
    for b in range(bits):
        for k in range(3):
            code |= ((g[:, k] >> b) & 1) << (3 * b + k)
    key = data_batch.to(torch.int64) * (1 << (3 * bits)) + code
    return torch.argsort(key, stable=True)


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


_ABI_CACHE: Dict[tuple, object] = {}   # one RCCL communicator per (device, group) and process; the key holds the group object
_SIDE_STREAMS: Dict[torch.device, "torch.cuda.Stream"] = {}


class _EventWork:
    """Completion of work enqueued on the communication stream: the compute stream waits for the event recorded behind it
    (stream-level dependency, no host wait; inside a HIP graph capture this becomes an edge between two branches)."""

    def __init__(self, ev, keep=(), dev=None):
        self.ev, self.keep, self.dev = ev, keep, dev      # keep: tensors the side stream reads / writes, alive until the join

    def wait(self):
        torch.cuda.current_stream(self.dev).wait_event(self.ev)
        self.keep = ()
        return True


class _Comm:
    """The exchange steps of SURVEY 8e (torch.distributed: RCCL on the GPU, gloo in the CPU tests; or the C-ABI transport).
    World size 1 short-circuits to local copies."""

    def __init__(self, group, stats: Optional[CommStats], emulate: bool = False):
        self.group, self.stats = group, stats
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.emulate = emulate
        # a transport without device support (gloo): device tensors are staged through host memory, synchronously
        self.host_staged = self.world > 1 and dist.get_backend(group) == "gloo"
        # Blocking collectives in program order (async_op=False / the compute stream: no overlap, nothing in flight behind a
        # kernel) are the DEFAULT while RCCL with more than one rank has never run on the hardware this was built on (ADVICE
        # round 2); FASTEGNN_SHARDED_SYNC=0 selects the asynchronous schedule: torch transport -> async_op=True, C-ABI
        # transport -> a second HIP stream forked / joined with events; each exchange is waited for at its first consumer.
        self.sync = os.environ.get("FASTEGNN_SHARDED_SYNC", "1") not in ("", "0")
        # FASTEGNN_COMM=abi: the data-path collectives go through the C ABI (fastegnn_comm_*: RCCL on a stream this process
        # chooses, capturable), created lazily on the first device tensor; plan-building exchanges stay on torch.distributed.
        # NOTE: that is a second RCCL communicator beside torch.distributed's; the two are never in flight
        # together -- allreduce_gradients runs after the backward on the compute stream, behind every ABI collective -- but
        # the combination is unverified on more than one rank: opt-in.
        self.use_abi = os.environ.get("FASTEGNN_COMM", "torch") == "abi" and not self.host_staged
        self._abi = None

    def abi(self, t):
        """The C-ABI communicator for tensor t's device, or None when this transport is not selected."""
        if not (self.use_abi and t.is_cuda and (self.world > 1 or self.emulate)):
            return None
        if self._abi is None:
            from .comm import AbiComm
            key = (t.device, self.group)
            self._abi = _ABI_CACHE.get(key)
            if self._abi is None:
                self._abi = _ABI_CACHE[key] = AbiComm(t.device, self.group)
        return self._abi

    # -- C-ABI transport: on the compute stream (sync) or forked onto the communication stream
    def _enqueue(self, t, fn, keep=()):
        if self.sync:
            fn()
            return _Done()
        cs = _SIDE_STREAMS.get(t.device)
        if cs is None:
            cs = _SIDE_STREAMS[t.device] = torch.cuda.Stream(t.device)
        cur = torch.cuda.current_stream(t.device)
        cs.wait_stream(cur)                      # everything enqueued so far precedes the collective
        with torch.cuda.stream(cs):
            fn()
            ev = torch.cuda.Event()
            ev.record(cs)
        return _EventWork(ev, keep, t.device)

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
            if t.is_cuda and not torch.cuda.is_current_stream_capturing():
                rec = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                rec[0].record()
                st.events.setdefault(name, []).append(rec)
        return rec

    def all_reduce(self, name, t):
        rec = self._note(name, t)
        a = self.abi(t)
        if a is not None:
            return _Timed(self._enqueue(t, lambda: a.all_reduce(t), (t,)), rec)
        if self.world == 1:
            return _Timed(_Done(), rec)
        if self.host_staged and t.is_cuda:
            return _Timed(self._staged(lambda o: dist.all_reduce(o, group=self.group), t), rec)
        return _Timed(dist.all_reduce(t, group=self.group, async_op=not self.sync) or _Done(), rec)

    def all_gather(self, name, out, inp):
        rec = self._note(name, out)
        if self.world == 1:
            out.copy_(inp)
            return _Timed(_Done(), rec)
        a = self.abi(out)
        if a is not None:
            return _Timed(self._enqueue(out, lambda: a.all_gather(out, inp), (out, inp)), rec)
        if self.host_staged and out.is_cuda:
            return _Timed(self._staged(lambda o, i: dist.all_gather_into_tensor(o, i, group=self.group), out, inp), rec)
        return _Timed(dist.all_gather_into_tensor(out, inp, group=self.group, async_op=not self.sync) or _Done(), rec)

    def all_to_all_v(self, name, out, inp, out_splits, in_splits):
        rec = self._note(name, out)
        if self.world == 1:
            return _Timed(_Done(), rec)
        a = self.abi(out)
        if a is not None:
            return _Timed(self._enqueue(out, lambda: a.all_to_all_v(out, inp.contiguous(), out_splits, in_splits), (out, inp)), rec)
        if self.host_staged and out.is_cuda:
            def fn(o, i):
                dist.all_to_all_single(o, i, out_splits, in_splits, group=self.group)
            return _Timed(self._staged(fn, out, inp.contiguous()), rec)
        w = dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group, async_op=not self.sync)
        return _Timed(w if w is not None else _Done(), rec)

    def halo_exchange(self, name, ghost_rows, table, send_ids, recv_splits, send_splits):
        """Forward table exchange: pack the rows the peers asked for (send_ids into `table`) and all-to-all-v them into
        `ghost_rows`.  C-ABI transport: pack kernel + grouped send / recv, both on the communication stream."""
        a = self.abi(ghost_rows)
        if a is not None and self.world > 1:
            rec = self._note(name, ghost_rows)
            box = []

            def fn():
                send = a.gather_rows(table, send_ids)
                box.append(send)
                a.all_to_all_v(ghost_rows, send, recv_splits, send_splits)
            return _Timed(self._enqueue(ghost_rows, fn, (ghost_rows, table, box)), rec)
        send = table.index_select(0, send_ids) if send_ids.numel() else table.new_zeros(0, table.size(1))
        return self.all_to_all_v(name, ghost_rows, send, recv_splits, send_splits)

    def scatter_add_rows(self, table, ids, rows):
        """table[ids] += rows (ids may repeat: a row asked for by several peers)"""
        a = self.abi(table)
        if a is not None:
            a.scatter_add_rows(table, ids, rows)
        else:
            table.index_add_(0, ids, rows)

    # -- one rank of a larger world in a single process (bench.py --emulate-world): the same pack / unpack kernels and byte
    #    counts, device copies in the place of the xGMI transfers; the ghost rows are filled with own rows (timing only)
    def emulated_exchange(self, name, ghost_rows, table, send_ids):
        rec = self._note(name, ghost_rows)
        a = self.abi(ghost_rows)
        n_g, n_own = ghost_rows.size(0), table.size(0)
        fill = (torch.arange(n_g, device=table.device) * 7919) % max(n_own, 1)

        def fn():
            send = a.gather_rows(table, send_ids) if a is not None else table.index_select(0, send_ids)
            ghost_rows.copy_(a.gather_rows(table, fill) if a is not None else table.index_select(0, fill))
            return send
        return _Timed(self._enqueue(ghost_rows, fn, (ghost_rows, table, fill)), rec)

    def emulated_return(self, name, back, send):
        rec = self._note(name, back)
        n = min(back.size(0), send.size(0))

        def fn():
            back.zero_()
            if n:
                back[:n].copy_(send[:n])
        return _Timed(self._enqueue(back, fn, (back, send)), rec)

    # plan-building exchanges (once per graph, blocking): int64 counts [W] and id lists
    def exchange_counts(self, counts):
        if self.world == 1:
            return counts.clone()
        out = torch.empty_like(counts)
        if self.host_staged and counts.is_cuda:
            h = counts.cpu()
            ho = torch.empty_like(h)
            dist.all_to_all_single(ho, h, group=self.group)
            return ho.to(counts.device)
        dist.all_to_all_single(out, counts, group=self.group)
        return out

    def exchange_ids(self, ids, out_splits, in_splits):
        out = torch.empty(sum(out_splits), dtype=ids.dtype, device=ids.device)
        if self.world == 1:
            return out
        if self.host_staged and ids.is_cuda:
            ho = torch.empty(sum(out_splits), dtype=ids.dtype)
            dist.all_to_all_single(ho, ids.cpu().contiguous(), out_splits, in_splits, group=self.group)
            return ho.to(ids.device)
        dist.all_to_all_single(out, ids.contiguous(), out_splits, in_splits, group=self.group)
        return out

    def reduce_scatter(self, name, out, inp):
        rec = self._note(name, inp)
        if self.world == 1:
            out.copy_(inp)
            return _Timed(_Done(), rec)
        a = self.abi(out)
        if a is not None:
            return _Timed(self._enqueue(out, lambda: a.reduce_scatter(out, inp), (out, inp)), rec)
        if self.host_staged and out.is_cuda:
            return _Timed(self._staged(lambda o, i: dist.reduce_scatter_tensor(o, i, group=self.group), out, inp), rec)
        return _Timed(dist.reduce_scatter_tensor(out, inp, group=self.group, async_op=not self.sync) or _Done(), rec)


class _LayerList(list):
    """The FASTEGNN_P_* ordered tensors of one layer.  The host array of their device pointers is built once and lives with the
    list (a stage call needs it; a layer makes 6 forward / 7 backward stage calls with the same list -- 38 data_ptr() calls each)."""

    _ptab = None

    def ptab(self):
        if self._ptab is None:
            self._ptab = _PtrTable(self)
            # no reference back to the list: a cycle would keep the gradient views alive past the backward, and autograd's
            # AccumulateGrad then CLONES them instead of adopting the slices of the one flat buffer (dist.allreduce_gradients
            # reduces that buffer in place)
            self._ptab.keep = None
        return self._ptab


def _layer_lists(spec: _Spec, params, i):
    return _LayerList(params[s] if s is not None else None for s in spec.layer_slots[i])


class _Part:
    """One launch range of the edge stage: rows [row0, row0 + nrows) of the rank, their edges as a sorted graph."""

    def __init__(self, graph, row0, nrows, halo, mask):
        # mask: (first, last) positions of the part's edges in the rank's edge list, or None (the whole list)
        self.graph, self.row0, self.nrows, self.halo, self.mask = graph, row0, nrows, halo, mask
        self.ea_sorted = None


_EDGE_ROW_KEYS = ("P", "QX", "x", "aggm", "aggx", "g_aggm", "g_aggx", "g_P", "g_xrow")


def _edge_view(t: Dict[str, torch.Tensor], part: _Part, **extra):
    """The layer's buffers as the edge stage sees them for one part: every row-indexed array starts at the part's first row
    (the part's graph numbers its rows from there), the source table and everything col-indexed stay whole."""
    v = dict(t)
    if part.row0 or part.nrows != t["P"].size(0):
        for k in _EDGE_ROW_KEYS:
            if k in v and v[k] is not None:
                v[k] = v[k][part.row0:part.row0 + part.nrows]
    v["ea_sorted"] = part.ea_sorted
    v.update(extra)
    return v


class _ShardedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, comm, spec: _Spec, plan: ShardPlan, parts, batch32, gptr, edge_attr, node_attr,
                node_feat, node_loc, node_vel, loc_mean, *params):
        # edge_attr / node_attr: this rank's edges / rows; differentiable like in the single-GPU module (their gradients
        # are accumulated by the edge / virtual backward kernels when asked for)
        for pt in parts:
            ea_p = edge_attr.detach() if edge_attr is not None else None
            if ea_p is not None and pt.mask is not None:
                ea_p = ea_p[pt.mask[0]:pt.mask[1]]
            pt.ea_sorted = pt.graph.permute(ea_p)
        node_attr = node_attr.detach().contiguous().float() if node_attr is not None else None
        N, B, Cn = plan.nloc, loc_mean.size(0), spec.C
        params = [p.detach() for p in params]
        node_feat, node_loc, node_vel, loc_mean = (t.detach().contiguous().float()
                                                   for t in (node_feat, node_loc, node_vel, loc_mean))
        g0 = parts[0].graph
        h = be.empty(N, H)
        be.embed_forward(node_feat, spec.nf, params[1], params[2], h)
        HvT = be.empty(B, Cn, H)
        be.virtual_init(params[0], B, Cn, HvT)
        x, Z = node_loc, loc_mean
        saved = []
        # Order of the collectives (identical on all ranks): xsum_0 | per layer: QX_i, [pools_i + xsum_{i+1}].  Each is
        # waited for at its first consumer; what runs in between does not depend on it:
        #   halo QX_i              behind  graph_post_{i-1} (deferred), graph_pre_i, edge_i over the INTERIOR rows
        #   all-reduce pools_i     behind  pack_{i+1}, node_pre_{i+1}
        pend = None     # (work, t, lp, b) of the previous layer: its graph_post is still to run
        xsum = be.empty(B, 4)
        be.stage("graph_xsum", spec, N, B, g0, dict(batch=batch32, x=x, xsum=xsum), _layer_lists(spec, params, 0))
        w_xsum = comm.all_reduce("xsum", xsum)
        nwp = (be.wpack_floats(Cn) + 3) // 4 * 4
        wpacks = be.empty(spec.n_layers, nwp)
        be.pack_all(spec, N, B, g0, [_layer_lists(spec, params, i) for i in range(spec.n_layers)], wpacks)
        for i in range(spec.n_layers):
            lp = _layer_lists(spec, params, i)
            b = dict(h=h, x=x, xsum=xsum, wpack=wpacks[i])
            b.update(be.carve(dict(P=(N, H), A=(N, H), svel=(N,), sgrav=(N,),
                                   Bc=(B, Cn, H), aggm=(N, H), npre=(N, H), aggx=(N, 3))))
            b["QX"], b["QX_src"] = plan.alloc_tables(be)           # own rows | the table the edge kernels gather from
            nV, nX = B * Cn * H, B * 3 * Cn
            last = i + 1 == spec.n_layers
            pools = be.empty(nV + nX + (0 if last else B * 4))     # poolV | poolX | xsum of the next layer: one all-reduce
            b["poolV"], b["poolX"] = pools[:nV].view(B, Cn, H), pools[nV:nV + nX].view(B, 3, Cn)
            b.update(h_out=be.empty(N, H), x_out=be.empty(N, 3), Z_out=be.empty(B, 3, Cn), HvT_out=be.empty(B, Cn, H))
            t = dict(batch=batch32, gptr=gptr, vel=node_vel, node_attr=node_attr, **b)
            be.stage("node_pre_forward", spec, N, B, g0, t, lp)
            w_qx = plan.exchange_forward(comm, b["QX"], b["QX_src"])
            if pend is not None:                                   # virtual state of this layer <- previous layer's pools
                pend[0].wait()
                be.stage("graph_post_forward", spec, N, B, g0, pend[1], pend[2])
                Z, HvT = pend[3]["Z_out"], pend[3]["HvT_out"]
                for k in ("aggx", "poolX", "h_out", "x_out", "Z_out", "HvT_out"):
                    del pend[3][k]
            else:
                w_xsum.wait()
            b["Z"], b["HvT"] = Z, HvT
            t["Z"], t["HvT"] = Z, HvT
            be.stage("graph_pre_forward", spec, N, B, g0, t, lp)
            waited = False
            for pt in parts:                                       # interior rows first: they read own rows only
                if pt.halo and not waited:
                    w_qx.wait()
                    waited = True
                be.stage("edge_forward", spec, pt.nrows, B, pt.graph, _edge_view(t, pt), lp)
            if not waited:
                w_qx.wait()
            be.stage("virt_forward", spec, N, B, g0, t, lp)
            if not last:                                           # x_out is final: the next layer's centroid sums ride along
                xsum = pools[nV + nX:].view(B, 4)
                be.stage("graph_xsum", spec, N, B, g0, dict(batch=batch32, x=b["x_out"], xsum=xsum), lp)
            pend = (comm.all_reduce("pools", pools), t, lp, b)
            saved.append(b)
            h, x = b["h_out"], b["x_out"]
        pend[0].wait()
        be.stage("graph_post_forward", spec, N, B, g0, pend[1], pend[2])
        Z = pend[3]["Z_out"]
        for k in ("aggx", "poolX", "h_out", "x_out", "Z_out", "HvT_out"):
            del pend[3][k]
        ctx.be, ctx.spec, ctx.plan, ctx.parts, ctx.saved = be, spec, plan, parts, saved
        ctx.comm = comm
        ctx.misc = (batch32, gptr, node_attr, node_feat, node_vel, params)
        ctx.ea_shape = edge_attr.shape if edge_attr is not None else None
        return x, Z

    @staticmethod
    def backward(ctx, g_loc, g_vloc):
        be, spec, plan, parts, saved = ctx.be, ctx.spec, ctx.plan, ctx.parts, ctx.saved
        comm = ctx.comm
        batch32, gptr, node_attr, node_feat, node_vel, params = ctx.misc
        rank = plan.rank
        N, B, Cn = plan.nloc, saved[0]["Z"].size(0), spec.C
        g0 = parts[0].graph
        E_max = max(pt.graph.E for pt in parts)
        det = bool(spec.flags & K.F_DETERMINISTIC)
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
        g_h = be.zeros(N, H)
        g_x = (g_loc if g_loc is not None else be.zeros(N, 3)).contiguous().float()
        g_Z = (g_vloc if g_vloc is not None else be.zeros(B, 3, Cn)).contiguous().float()
        g_HvT = be.zeros(B, Cn, H)
        g_vel = be.zeros(N, 3)
        sc = be.carve(dict(g_poolV=(B, Cn, H), g_poolX=(B, 3, Cn), g_xbar=(B, 4), g_A=(N, H), g_P=(N, H),
                           g_aggm=(N, H), g_aggx=(N, 3), g_svel=(N,), g_sgrav=(N,), g_QXe=(max(E_max, 1) if det else 1, K.QX_LD),
                           g_xrow=(N, 3), wg_edge=(be.wg_edge_floats(E_max),), wg_virt=(be.wg_virt_floats(N, Cn, spec.flags),),
                           wg_node=(be.wg_node_floats(N, B, Cn),), wg_slab=(be.wg_slab_floats(),)))
        nV = B * Cn * H
        gpools = be.empty(nV + B * 3 * Cn)                                 # g_Bc | g_Zp adjacent: one all-reduce
        sc["g_Bc"], sc["g_Zp"] = gpools[:nV].view(B, Cn, H), gpools[nV:].view(B, 3, Cn)
        sc["g_QX_src"], sc["g_QX"] = plan.alloc_grad_tables(be)
        has_ea = ctx.ea_shape is not None and ctx.ea_shape[1] > 0
        want_ea = ctx.needs_input_grad[7] and has_ea
        want_na = ctx.needs_input_grad[8] and node_attr is not None
        g_ea_parts = [be.zeros(max(pt.graph.E, 1), spec.ea) if want_ea else None for pt in parts]
        if want_na:
            sc["g_node_attr"] = be.zeros(N, spec.na)
        # backward order of the parts: the rows that feed ghost columns first, so that their share of g_QX_src travels while
        # the interior rows are computed
        bparts = sorted(range(len(parts)), key=lambda k: not parts[k].halo)
        for i in reversed(range(spec.n_layers)):
            b = saved[i]
            lp = _layer_lists(spec, params, i)
            lg = _layer_lists(spec, grads, i)
            # the per-graph stages run on every rank (their outputs are needed everywhere); only rank 0 contracts their
            # weight gradients -- the other ranks pass an empty gradient table and queue no job for them
            lr = lg if rank == 0 else [None] * len(lg)
            out = dict(g_h=be.empty(N, H), g_x=be.empty(N, 3), g_Z=be.empty(B, 3, Cn), g_HvT=be.empty(B, Cn, H))
            t = dict(batch=batch32, gptr=gptr, vel=node_vel, node_attr=node_attr,
                     g_h_out=g_h, g_x_out=g_x, g_Z_out=g_Z, g_HvT_out=g_HvT, g_vel=g_vel, **b, **out, **sc)
            wb = be.wgrad_open(spec, N, B, g0, t, lp)              # one contraction + one reduction launch for the layer
            t["wgrad_batch"] = wb
            try:
                be.stage("graph_post_backward", spec, N, B, g0, t, lp, lr)
                be.stage("virt_backward", spec, N, B, g0, t, lp, lg)
                w_pools = comm.all_reduce("g_pools", gpools)       # behind the edge backward
                w_qx = _Done()
                n_halo = max(sum(1 for pt in parts if pt.halo), 1)
                for n_done, k in enumerate(bparts):
                    pt = parts[k]
                    extra = dict(g_ea_sorted=g_ea_parts[k]) if want_ea else {}
                    acc = K.F_GQX_ACCUM if n_done > 0 else 0      # (a second part exists in the atomic mode only)
                    be.stage("edge_backward", spec, pt.nrows, B, pt.graph, _edge_view(t, pt, **extra), lp, lg, flags=acc)
                    be.stage("edge_col_reduce", spec, pt.nrows, B, pt.graph, _edge_view(t, pt), lp, lg)
                    if n_done + 1 == n_halo:   # everything that scatters into ghost rows has run: send them home now
                        w_qx = plan.exchange_backward(comm, sc["g_QX_src"], sc["g_QX"], det)
                w_pools.wait()
                be.stage("graph_pre_backward", spec, N, B, g0, t, lp, lr)
                w_qx.wait()
                be.stage("node_pre_backward", spec, N, B, g0, t, lp, lg)
            finally:
                be.wgrad_close(wb)
            g_h, g_x, g_Z, g_HvT = out["g_h"], out["g_x"], out["g_Z"], out["g_HvT"]
            saved[i] = None
        be.virtual_init_backward(g_HvT, B, Cn, grads[0] if rank == 0 else be.zeros(*params[0].shape))
        g_nf = torch.empty_like(node_feat) if ctx.needs_input_grad[9] else None
        be.embed_backward(node_feat, g_h, spec.nf, params[1], grads[1], grads[2], g_nf)
        # The last layer's node_mlp / node_mlp_virtual feed nothing: the reference's autograd (and the single-GPU module)
        # leave their .grad None and torch.optim.Adam / FusedAdam then skip them; the kernels wrote zeros.
        last = spec.n_layers - 1
        for s_, suffix in zip(spec.layer_slots[last], K.PARAM_SLOTS):
            if s_ is not None and suffix.startswith(("node_mlp.", "node_mlp_virtual.")) and not (spec.flags & K.F_RF):
                grads[s_] = None
        g_ea = None
        if want_ea:   # back to this rank's edge order: sorted edge k of a part is the part's input edge perm[k]
            g_ea = be.zeros(*ctx.ea_shape)
            for pt, gp in zip(parts, g_ea_parts):
                E = pt.graph.E
                if E == 0:
                    continue
                un = torch.empty_like(gp[:E])
                un.index_copy_(0, pt.graph.perm[:E].long(), gp[:E])
                if pt.mask is None:
                    g_ea = un
                else:
                    g_ea[pt.mask[0]:pt.mask[1]] = un
        return (None,) * 7 + (g_ea, sc["g_node_attr"] if want_na else None, g_nf, g_x, g_vel, g_Z, *grads)


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

    def __init__(self, model: FastEGNN, group=None, backend=None, stats: Optional[CommStats] = None,
                 exchange: Optional[str] = None, emulate: Optional[tuple] = None):
        """exchange: "halo" (ghost rows only, all-to-all-v; default) or "allgather" (the whole table); the environment
        variable FASTEGNN_SHARDED_EXCHANGE overrides the default.  emulate=(world, rank): this single process plays ONE
        rank of a larger world (bench.py --emulate-world): its share of the rows and edges, the pack / unpack kernels and
        byte counts of its halo, device copies in the place of the transfers -- timing of a shard's compute and fixed
        costs on one GPU; the ghost rows hold stand-in data, so the outputs mean nothing."""
        super().__init__()
        if getattr(model, "_wide", False):
            raise NotImplementedError("ShardedFastEGNN: hidden_nf > 64 (the unfused wide path) runs on one GPU only")
        self.exchange = exchange or os.environ.get("FASTEGNN_SHARDED_EXCHANGE", "halo")
        if self.exchange not in ("halo", "allgather"):
            raise ValueError("ShardedFastEGNN: exchange must be 'halo' or 'allgather'")
        if emulate is not None and (self.exchange != "halo" or dist.is_initialized() and dist.get_world_size(group) > 1):
            raise ValueError("ShardedFastEGNN: emulate=(world, rank) needs the halo exchange and a single process")
        self.model = model
        self.group = group
        self.backend = backend
        self.stats = stats
        self.emulate = emulate
        self.plan: Optional[ShardPlan] = None
        self._comm: Optional[_Comm] = None

    def _world_rank(self):
        if self.emulate is not None:
            return self.emulate
        if not dist.is_initialized():
            return 1, 0
        return dist.get_world_size(self.group), dist.get_rank(self.group)

    def comm(self) -> _Comm:
        if self._comm is None or self._comm.stats is not self.stats:
            self._comm = _Comm(self.group, self.stats, emulate=self.emulate is not None)
        return self._comm

    def shard_inputs(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None,
                     node_attr=None, reorder: bool = False, split: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """This rank's share of a batch: its node rows, the edges aggregating into them, the replicated per-graph
        tensors.  Collective when the exchange is "halo" (the ranks tell each other which rows they need).
        reorder=True first sorts the nodes of every graph along a Morton curve (`morton_order`), so that a rank owns a
        compact region of space; `plan.node_ids` / `plan.rows()` map the caller's node order to this rank's rows.
        split: order the rank's rows [interior | boundary] and run the edge stage as two launches (default: with the
        asynchronous schedule, FASTEGNN_SHARDED_SYNC=0, where the halo then travels behind the interior rows; never with the
        deterministic backward, whose col-keyed sum runs over ONE graph; FASTEGNN_SHARDED_SPLIT=0/1 overrides)."""
        world, rank = self._world_rank()
        N = node_loc.size(0)
        order = None
        if reorder:
            order = morton_order(node_loc.detach(), data_batch)
            inv = torch.empty_like(order)
            inv[order] = torch.arange(N, device=order.device)
            edge_index = inv[edge_index]
        plan = (HaloPlan if self.exchange == "halo" else ShardPlan)(N, world, rank, order)
        # (edge_attr / node_attr keep their autograd link: the selection below is differentiable, so a gradient computed
        # for this rank's edges / rows flows back into the caller's full tensors -- zero where another rank owns the edge)
        ei, ea = plan.edges(edge_index, edge_attr)
        db = data_batch if order is None else data_batch[order]
        if plan.mode == "halo":
            if split is None:
                # Two launch ranges pay when the halo travels BEHIND the interior rows (the asynchronous schedule); with the
                # collectives in program order they only cost a second CSR build and eight more launches per step (measured on
                # an emulated rank of eight: 3.46 against 3.1x ms) -- so: on with FASTEGNN_SHARDED_SYNC=0, off otherwise, and
                # FASTEGNN_SHARDED_SPLIT=0/1 overrides.  Never with the deterministic backward (one col-keyed sum per graph).
                env = os.environ.get("FASTEGNN_SHARDED_SPLIT", "")
                want = (env not in ("", "0")) if env != "" else not self.comm().sync
                split = want and not bool(getattr(self.model, "deterministic", False))
            want = None
            if self.emulate is not None:
                # the rows of this rank its peers' edges read, grouped by peer (a real world learns them in plan.build)
                rows, cols = edge_index[0], edge_index[1]
                m = ((rows < plan.n0) | (rows >= plan.n1)) & (cols >= plan.n0) & (cols < plan.n1)
                peer = torch.div(rows[m], plan.Npad, rounding_mode="floor")
                want = torch.unique(peer * N + cols[m]) % N          # (peer, col) pairs, ascending by peer then col
            ei = plan.build(ei, self.comm(), db[plan.n0:plan.n1], split, peers_want=want)
            if plan.edge_perm is not None and ea is not None:
                ea = ea[plan.edge_perm]
        return dict(plan=plan, node_feat=plan.rows(node_feat), node_loc=plan.rows(node_loc), node_vel=plan.rows(node_vel),
                    edge_index=ei, edge_attr=ea, data_batch=plan.rows(data_batch), loc_mean=loc_mean,
                    node_attr=plan.rows(node_attr).float() if node_attr is not None else None)

    def forward_local(self, local: Dict[str, torch.Tensor]):
        m = self.model
        plan: ShardPlan = local["plan"]
        if m._spec is None:
            m._spec = _Spec(m)
            pidx = m._param_index
            m._plist = [pidx[n] for n in m._spec.names]
        # the range guard of the wrapped module decides the build (fastegnn_amd.model.RangeGuard); a caller-supplied backend
        # (the CPU stage backend of the gloo tests) is used as it is
        guard = m._range
        dev = local["node_loc"].device
        guarded = self.backend is None and not (m._spec.flags & K.F_BF16)
        world, _ = self._world_rank()
        group = (self.group if self.group is not None else dist.group.WORLD) if (dist.is_initialized() and self.emulate is None and world > 1) else None
        if guarded and not guard.wide and group is None:
            guard.poll("ShardedFastEGNN", m._plist)     # one process: the deferred check of fastegnn_amd.model.RangeGuard (no synchronisation)
        be = self.backend or HipBackend(dev, act=m._spec.act_kind != K.ACT_SILU, wide=guard.wide)
        spec = m._spec
        csc = bool(m._spec.flags & K.F_DETERMINISTIC)
        if csc and len(plan.parts) > 1:
            raise RuntimeError("ShardedFastEGNN: the deterministic backward needs an unsplit plan (shard_inputs(split=False))")
        ei = local["edge_index"]
        parts, e0 = [], 0
        counts = plan.edge_counts if plan.edge_counts is not None else [ei.size(1)]
        for (row0, nrows, halo), ne in zip(plan.parts, counts):
            sl = None if len(plan.parts) == 1 else (e0, e0 + ne)      # the part's edges: a slice of the rank's edge list
            ei_p = ei if sl is None else ei[:, sl[0]:sl[1]].contiguous()
            parts.append(_Part(be.build_graph(ei_p, nrows, plan.n_src, plan.n0 + row0, csc=csc), row0, nrows, halo, sl))
            e0 += ne
        B = local["loc_mean"].size(0)
        batch32, gptr = be.build_batch(local["data_batch"], plan.nloc, B)
        self.plan = plan
        ea = local["edge_attr"]
        if ea is not None and ea.size(1) == 0:
            ea = None
        plist = m._plist
        if m.hidden_nf < K.H:
            plist = be.pad_params(spec.names, m.hidden_nf, spec.C, bool(spec.flags & K.F_RF), plist)
        def run(be_):
            return _ShardedFunction.apply(be_, self.comm(), spec, plan, parts, batch32, gptr, ea,
                                          local["node_attr"], local["node_feat"], local["node_loc"], local["node_vel"],
                                          local["loc_mean"], *plist)
        out = run(be)
        if guarded and not guard.wide:
            # every rank checks ITS rows and the replicated virtual coordinates.  Several ranks: the builds must agree (halo rows carry
            # Q in the build's units), so the decision is taken over all ranks after every eager forward -- one stream synchronisation
            # and one 8-byte all-reduce, beside the blocking collectives this transport already has.  One process: deferred, as FastEGNN.
            guard.launch(be.lib, (out[0], out[1]), (local["node_loc"], local["node_vel"]))
            if (group is not None or guard.mode == "sync") and guard.sync_and_poll(dev, "ShardedFastEGNN", m._plist, group):
                out = run(HipBackend(dev, act=spec.act_kind != K.ACT_SILU, wide=True))
        return out

    def forward(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None,
                node_attr=None):
        return self.forward_local(self.shard_inputs(node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean,
                                                    edge_attr, node_attr))
