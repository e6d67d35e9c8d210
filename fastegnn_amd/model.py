"""Drop-in ``FastEGNN`` nn.Module backed by the gfx950 HIP library.

Mirrors the reference module surface (``/root/reference/models/FastEGNN.py:226-276``):
same class name (the reference harness dispatches on ``model.__class__.__name__``,
``utils/train.py:51``), same constructor and ``forward`` signatures, same ``state_dict``
keys/shapes and the same parameter-construction order (so a given ``torch.manual_seed``
yields the same initial weights).  The arithmetic runs in ``libfastegnn_hip.so`` through the
C ABI of ``include/fastegnn_hip.h``; there is no CPU path -- CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib as K

H = K.H


# ------------------------------------------------------------------------------------------
# parameter containers (never called: they only give the reference's state_dict layout)
# ------------------------------------------------------------------------------------------
class E_GCL_vel(nn.Module):
    """Parameter holder of one layer; construction order follows models/FastEGNN.py:27-99."""

    def __init__(self, hidden_nf, node_attr_nf, edge_attr_nf, virtual_channels, act_fn, attention, tanh, gravity):
        super().__init__()
        Hn, Cn = hidden_nf, virtual_channels
        self.edge_mlp = nn.Sequential(nn.Linear(2 * Hn + 1 + edge_attr_nf, Hn), act_fn, nn.Linear(Hn, Hn), act_fn)
        self.edge_mlp_virtual = nn.Sequential(nn.Linear(2 * Hn + 1 + Cn, Hn), act_fn, nn.Linear(Hn, Hn), act_fn)
        if attention:
            self.att_mlp = nn.Sequential(nn.Linear(Hn, 1), nn.Sigmoid())
            self.att_mlp_virtual = nn.Sequential(nn.Linear(Hn, 1), nn.Sigmoid())

        def coord_mlp():
            last = nn.Linear(Hn, 1, bias=False)
            torch.nn.init.xavier_uniform_(last.weight, gain=0.001)
            mods = [nn.Linear(Hn, Hn), act_fn, last]
            if tanh:
                mods.append(nn.Tanh())
            return nn.Sequential(*mods)

        self.coord_mlp_r = coord_mlp()
        self.coord_mlp_r_virtual = coord_mlp()
        self.coord_mlp_v_virtual = coord_mlp()
        self.coord_mlp_vel = nn.Sequential(nn.Linear(Hn, Hn), act_fn, nn.Linear(Hn, 1))
        if gravity is not None:
            self.gravity_mlp = nn.Sequential(nn.Linear(Hn, Hn), act_fn, nn.Linear(Hn, 1))
        self.node_mlp = nn.Sequential(nn.Linear(Hn + Hn + Cn * Hn + node_attr_nf, Hn), act_fn, nn.Linear(Hn, Hn))
        self.node_mlp_virtual = nn.Sequential(nn.Linear(Hn + Hn, Hn), act_fn, nn.Linear(Hn, Hn))

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("E_GCL_vel is evaluated inside FastEGNN.forward by the HIP library")


# FASTEGNN_DEBUG_CHECKS=1: validate the index inputs on every forward (host sync).  The kernels trust them: an
# out-of-range col is an out-of-bounds gather, an unsorted data_batch breaks the graph pointer search; the
# reference raises an index error in these cases.
_DEBUG_CHECKS = os.environ.get("FASTEGNN_DEBUG_CHECKS", "0") not in ("", "0")


class RangeGuard:
    """Automatic wide-range fallback of one module (FastEGNN / FastRF / EGNN / ShardedFastEGNN) -- WITHOUT a host synchronisation.

    The default library multiplies on 2-part fp16 splits: a hidden activation or a [64,64] weight beyond 65 504 overflows there
    and the outputs turn non-finite, where the reference's plain fp32 (models/FastEGNN.py:102-119) stays finite.  Every forward on
    the f16x2 build is therefore followed by ONE capturable launch (fastegnn_check_finite) that writes 1 into a pinned,
    host-MAPPED word when an output is Inf / NaN (and into a second word when an INPUT already was).  Nothing waits for it: the
    host polls the words with plain loads -- at the start of the next forward, at the start of the backward, in
    ``fastegnn_amd.train.train_step`` -- and only a set word does anything (round 6; the reference's one synchronisation,
    ``data_batch[-1].item()`` at models/FastEGNN.py:267, is not needed here either).  A set word switches the module to the
    wide-range build (libfastegnn_hip_x3.so / _act_x3.so) with one warning; the forward that overflowed has returned non-finite
    outputs by then, its backward hands ZERO parameter gradients to the optimizer (fastegnn_zero_if_flagged: the device reads the
    same word, so this does not depend on when the host looks) and the step is lost, like a skipped step of a loss scaler.
    Non-finite INPUTS or PARAMETERS do not switch the build (the wide-range build cannot repair them): they warn.
    HIP graphs captured before the switch keep replaying the f16x2 kernels and must be re-captured; the guard launch inside them
    keeps writing the words, so ``poll()`` sees an overflow of a replay too.

    ``mode``: "deferred" (default, above) or "sync" (``FASTEGNN_RANGE_CHECK=sync`` / ``module.range_check = "sync"``): the eager
    forward waits for its own guard launch and re-runs the call on the wide-range build before it returns -- what round 5 did,
    at one stream synchronisation per forward.  FASTEGNN_WIDE_RANGE=1 starts on the wide-range build, =0 pins the f16x2 build and
    turns the overflow into a FloatingPointError (at the poll that sees it); FASTEGNN_RANGE_CHECK=off queues no guard launch at all."""

    OUT, IN = 0, 1

    def __init__(self):
        self.forced = K.WIDE_RANGE            # None: automatic
        self.wide = bool(K.WIDE_RANGE)        # the build in use
        rc = os.environ.get("FASTEGNN_RANGE_CHECK", "deferred")
        self.mode = rc if rc in ("sync", "off") else "deferred"   # "off": no guard launches at all (the caller vouches for the operand range)
        self._words = None                    # ctypes int32[2], host-mapped (fastegnn_host_words_alloc)
        self.warned = False
        self.warned_inputs = False
        self.launched = False                 # a guard launch has been queued since the last poll

    def words(self):
        """the host-mapped words; allocated on first use (a host allocation: legal outside AND inside a stream capture)"""
        if self._words is None:
            p = C.POINTER(C.c_int32)()
            rc = K.lib().fastegnn_host_words_alloc(2, C.byref(p))
            if rc != 0 and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("fastegnn_amd: the range guard's host-mapped words could not be allocated inside a stream capture; "
                                   "run one eager forward of the module before capturing it into a HIP graph")
            K.check(rc, "fastegnn_host_words_alloc")
            self._words = p
        return self._words

    def word_ptr(self, i: int = 0) -> C.c_void_p:
        return C.c_void_p(C.addressof(self.words().contents) + 4 * i)

    def __del__(self):
        try:
            if self._words is not None:
                K.lib().fastegnn_host_words_free(self._words)
        except Exception:
            pass

    def launch(self, lib, outs, ins=()):
        """queue the check of up to two fp32 output tensors (word OUT) and up to two input tensors (word IN) on the current stream"""
        if self.mode == "off":
            return
        def two(ts):
            ts = [t.detach() for t in ts if t is not None and t.dtype == torch.float32 and t.is_contiguous() and t.numel()]
            return (ts + [None, None])[:2]
        dev = outs[0].device
        a, b = two(outs)
        K.check(lib.fastegnn_check_finite(K.ptr(a), a.numel() if a is not None else 0, K.ptr(b), b.numel() if b is not None else 0,
                                          self.word_ptr(self.OUT), _stream(dev)), "fastegnn_check_finite")
        a, b = two(ins)
        if a is not None:
            K.check(lib.fastegnn_check_finite(K.ptr(a), a.numel(), K.ptr(b), b.numel() if b is not None else 0,
                                              self.word_ptr(self.IN), _stream(dev)), "fastegnn_check_finite")
        self.launched = True

    def zero_if_flagged(self, lib, buf: torch.Tensor):
        """device side: buf = 0 if the OUT word is set (the parameter gradients of a backward whose forward ran on the f16x2 build)"""
        if self._words is None:
            return
        K.check(lib.fastegnn_zero_if_flagged(K.ptr(buf), buf.numel(), self.word_ptr(self.OUT), _stream(buf.device)),
                "fastegnn_zero_if_flagged")

    def peek(self) -> bool:
        """has a guard launch that already ran seen non-finite outputs?  (plain host loads; no synchronisation)"""
        return not self.wide and self._words is not None and self._words[self.OUT] != 0

    def poll(self, who: str, params=None, group=None, why: str = "an earlier forward pass") -> bool:
        """Look at the words (no synchronisation) and act on a set one.  -> True when the module was switched to the wide-range
        build by this call.  `group`: a torch.distributed group -- the decision is then taken over all its ranks (one small
        all-reduce of the two words: the sharded caller, whose builds must agree because halo rows carry build-specific units)."""
        if self.wide or self._words is None:
            return False
        w = self._words
        out_bad, in_bad = int(w[self.OUT]), int(w[self.IN])
        if group is not None:
            import torch.distributed as dist
            t = torch.tensor([out_bad, in_bad], dtype=torch.int32)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            out_bad, in_bad = (int(v) for v in t.tolist())
        if not out_bad:
            return False
        self.launched = False
        import warnings

        def decline():
            # the words are cleared only when the build stays: after a switch they are dead (no guard launch on the wide-range
            # build), and the backward of the pass that overflowed still reads OUT on the device (zero_if_flagged)
            w[self.OUT] = 0
            w[self.IN] = 0
            return False
        if in_bad:
            if not self.warned_inputs:
                warnings.warn(f"fastegnn_amd.{who}: {why} received non-finite INPUTS (its outputs are non-finite on any build); "
                              "the arithmetic build is unchanged", RuntimeWarning, stacklevel=3)
                self.warned_inputs = True
            return decline()
        if params is not None:   # (rare path: one reduction + synchronisation per parameter list)
            bad = [i for i, p in enumerate(params) if p is not None and not bool(torch.isfinite(p.detach()).all())]
            if bad:
                if not self.warned_inputs:
                    warnings.warn(f"fastegnn_amd.{who}: {len(bad)} parameter tensors hold non-finite values (a diverged or "
                                  "poisoned optimizer step); the arithmetic build is unchanged", RuntimeWarning, stacklevel=3)
                    self.warned_inputs = True
                return decline()
        self.switch(who, why)
        return True

    def switch(self, who: str, why: str = "a forward pass"):
        if self.forced is False:
            raise FloatingPointError(
                f"fastegnn_amd.{who}: {why} produced non-finite outputs on the f16x2 build (operands beyond 65 504) and "
                "FASTEGNN_WIDE_RANGE=0 pins that build; unset it (automatic fallback) or set FASTEGNN_WIDE_RANGE=1")
        if K.SAFE_WAITS:
            raise FloatingPointError(f"fastegnn_amd.{who}: {why} left the fp16 operand range; the FASTEGNN_SAFE_WAITS=1 diagnostic "
                                     "build has no wide-range form")
        self.wide = True
        if not self.warned:
            import warnings
            warnings.warn(f"fastegnn_amd.{who}: {why} left the operand range of the default f16x2 arithmetic (|x| > 65 504) and "
                          "returned non-finite outputs; this module now runs on the wide-range build (bf16x3 products, fp32's "
                          "exponent range, ~8 % slower).  The parameter gradients of that pass were zeroed on the device; HIP "
                          "graphs captured from this module before now still replay the f16x2 kernels and must be re-captured.  "
                          "FASTEGNN_WIDE_RANGE=1 selects the wide-range build from the start, FASTEGNN_RANGE_CHECK=sync re-runs "
                          "an overflowing call before it returns (one stream synchronisation per forward).", RuntimeWarning,
                          stacklevel=4)
            self.warned = True

    def sync_and_poll(self, dev, who: str, params=None, group=None) -> bool:
        """the "sync" mode / the sharded caller: wait for this forward's guard launch, then poll"""
        if torch.cuda.is_current_stream_capturing():
            return False
        torch.cuda.current_stream(dev).synchronize()
        return self.poll(who, params, group, why="a forward pass")


# ------------------------------------------------------------------------------------------
# sorted graph handle
# ------------------------------------------------------------------------------------------
_CSR_TMP: Dict[torch.device, torch.Tensor] = {}


class SortedGraph:
    """Device-side CSR + col-keyed index of a COO ``edge_index`` (fastegnn_build_csr)."""

    def __init__(self, edge_index: torch.Tensor, n_rows: int, n_src: Optional[int] = None, row_begin: int = 0,
                 csc: bool = True):
        """``csc=False`` skips the col-keyed index (a second radix sort): only the deterministic backward
        (FASTEGNN_F_DETERMINISTIC) reads it."""
        assert edge_index.is_cuda and edge_index.dtype == torch.int64 and edge_index.dim() == 2
        edge_index = edge_index.contiguous()
        dev = edge_index.device
        E = edge_index.size(1)
        n_src = n_rows if n_src is None else n_src
        if _DEBUG_CHECKS and E:
            # the radix sorts cover only the bits an id of the given range can have: an id outside its range would be mis-sorted
            # silently -- halo-remapped columns of the sharded path make that more plausible than it used to be
            r_lo, r_hi = int(edge_index[0].min()) - row_begin, int(edge_index[0].max()) - row_begin
            c_lo, c_hi = int(edge_index[1].min()), int(edge_index[1].max())
            if r_lo < 0 or r_hi >= n_rows or c_lo < 0 or c_hi >= n_src:
                raise IndexError(f"fastegnn_amd: edge_index rows span [{r_lo}, {r_hi}] of {n_rows} rows, columns [{c_lo}, {c_hi}] "
                                 f"of a {n_src}-row source table")
        i32 = dict(dtype=torch.int32, device=dev)
        self.n_rows, self.n_src, self.E = n_rows, n_src, E
        self.rowptr = torch.empty(n_rows + 1, **i32)
        self.erow = torch.empty(max(E, 1), **i32)
        self.col = torch.empty(max(E, 1), **i32)
        self.perm = torch.empty(max(E, 1), **i32)
        self.cscptr = torch.empty(n_src + 1, **i32) if csc else None
        self.csc_eid = torch.empty(max(E, 1), **i32) if csc else None
        L = K.lib()
        self.chunk_row = torch.empty(L.fastegnn_chunk_rows(E), **i32)
        nbytes = L.fastegnn_csr_tmp_bytes(E, n_rows, n_src)
        # the radix-sort workspace (28 B per edge) is kept per device and reused by later builds on the same
        # stream: at 20 M edges a fresh 0.6 GB block per step sporadically costs a device malloc (~1 s)
        tmp = _CSR_TMP.get(dev)
        if tmp is None or tmp.numel() < nbytes:
            tmp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _CSR_TMP[dev] = tmp
        nch = C.c_int32(0)
        st = torch.cuda.current_stream(dev).cuda_stream
        K.check(L.fastegnn_build_csr(K.ptr(edge_index), E, row_begin, n_rows, n_src, K.ptr(self.rowptr),
                                     K.ptr(self.erow), K.ptr(self.col), K.ptr(self.perm), K.ptr(self.cscptr),
                                     K.ptr(self.csc_eid), K.ptr(self.chunk_row), C.byref(nch), K.ptr(tmp), nbytes,
                                     C.c_void_p(st)), "fastegnn_build_csr")
        self.n_chunks = nch.value
        self._key = (edge_index.data_ptr(), E, edge_index._version, n_rows, n_src, row_begin)

    def struct(self) -> K.GraphT:
        g = K.GraphT()
        g.n_rows, g.n_src, g.n_edges, g.n_chunks = self.n_rows, self.n_src, self.E, self.n_chunks
        for n in ("rowptr", "erow", "col", "perm", "cscptr", "csc_eid", "chunk_row"):
            t = getattr(self, n)
            setattr(g, n, t.data_ptr() if t is not None else None)
        return g

    def permute(self, edge_attr: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if edge_attr is None or edge_attr.size(1) == 0:
            return None
        edge_attr = edge_attr.contiguous().float()
        out = torch.empty_like(edge_attr)
        st = torch.cuda.current_stream(edge_attr.device).cuda_stream
        K.check(K.lib().fastegnn_permute_rows(K.ptr(edge_attr), K.ptr(self.perm), self.E, edge_attr.size(1),
                                              K.ptr(out), C.c_void_p(st)), "fastegnn_permute_rows")
        return out


# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------
def _check_indices(edge_index, data_batch, N, B):
    if isinstance(edge_index, torch.Tensor) and edge_index.numel():
        lo, hi = int(edge_index.min()), int(edge_index.max())
        if lo < 0 or hi >= N:
            raise IndexError(f"fastegnn_amd: edge_index values span [{lo}, {hi}] but there are {N} nodes")
    if data_batch.numel():
        if bool((data_batch[1:] < data_batch[:-1]).any()):
            raise ValueError("fastegnn_amd: data_batch must be sorted ascending (as PyG collate emits it)")
        if int(data_batch[0]) < 0 or int(data_batch[-1]) >= B:
            raise IndexError(f"fastegnn_amd: data_batch values exceed the {B} graphs of loc_mean")


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _PtrTable:
    """HOST array of device pointers in FASTEGNN_P_* order for one layer."""

    def __init__(self, tensors: List[Optional[torch.Tensor]]):
        self.keep = tensors
        self.arr = (C.c_void_p * K.P_COUNT)(*[(t.data_ptr() if t is not None else None) for t in tensors])

    def addr(self):
        return C.cast(self.arr, C.c_void_p)


def _carve(dev, shapes: Dict[str, tuple]) -> Dict[str, torch.Tensor]:
    """One device allocation, carved into 16-byte aligned fp32 views (fewer allocator calls per step)."""
    sizes = {}
    for k, s in shapes.items():
        n = 1
        for d in s:
            n *= d
        sizes[k] = (n + 3) // 4 * 4
    flat = torch.empty(sum(sizes.values()), dtype=torch.float32, device=dev)
    out, off = {}, 0
    for k, s in shapes.items():
        n = 1
        for d in s:
            n *= d
        out[k] = flat[off:off + n].view(*s)
        off += sizes[k]
    return out


def _fill(layer: K.LayerT, **tensors):
    for name, t in tensors.items():
        setattr(layer, name, t.data_ptr() if isinstance(t, torch.Tensor) else t)


class _Spec:
    """Static description shared by forward and backward."""

    def __init__(self, model: "FastEGNN", deterministic: Optional[bool] = None):
        self.C = model.virtual_channels
        self.n_layers = model.n_layers
        self.ea = model.edge_attr_nf
        self.na = model.node_attr_nf
        self.nf = model.node_feat_nf
        flags = 0
        flags |= K.F_ATTENTION if model.attention else 0
        flags |= K.F_NORMALIZE if model.normalize else 0
        flags |= K.F_TANH if model.tanh else 0
        flags |= K.F_RESIDUAL if model.residual else 0
        flags |= K.F_GRAVITY if model.gravity is not None else 0
        flags |= K.F_BF16 if getattr(model, "mlp_dtype", torch.float32) == torch.bfloat16 else 0
        det = getattr(model, "deterministic", False) if deterministic is None else deterministic
        flags |= K.F_DETERMINISTIC if det else 0
        self.act_kind, self.act_param = getattr(model, "_act", (K.ACT_SILU, 0.0))
        flags |= self.act_kind << K.F_ACT_SHIFT
        self.flags = flags | model._extra_flags
        self.wide = False      # the build the NEXT call runs on (set by the module from its RangeGuard before every call)
        self.guard = None      # the module's RangeGuard while calls run on the f16x2 build (the backward polls it and zeroes by it)
        self.gravity = [float(v) for v in model.gravity] if model.gravity is not None else [0.0, 0.0, 0.0]
        # parameter order handed to the autograd function
        self.names = ["virtual_node_feat", "embedding_in.weight", "embedding_in.bias"]
        self.layer_slots: List[List[Optional[int]]] = []
        pidx = model._param_index
        for i in range(model.n_layers):
            slots = []
            for suffix in K.PARAM_SLOTS:
                key = f"gcl_{i}.{suffix}"
                if key in pidx:
                    slots.append(len(self.names))
                    self.names.append(key)
                else:
                    slots.append(None)
            self.layer_slots.append(slots)


def _new_layer(spec: _Spec, N: int, B: int, graph: SortedGraph) -> K.LayerT:
    L = K.LayerT()
    L.N, L.B, L.C, L.ea, L.na, L.flags = N, B, spec.C, spec.ea, spec.na, spec.flags
    L.gravity = (C.c_float * 3)(*spec.gravity)
    L.epsilon = 1e-8
    L.act_param = getattr(spec, "act_param", 0.0)
    L.graph = graph.struct()
    return L


# ------------------------------------------------------------------------------------------
# hidden_nf < 64: zero padding to the kernels' 64-wide tiles
# ------------------------------------------------------------------------------------------
_HIDDEN_OUT = ("edge_mlp.0", "edge_mlp.2", "edge_mlp_virtual.0", "edge_mlp_virtual.2", "coord_mlp_r.0",
               "coord_mlp_r_virtual.0", "coord_mlp_v_virtual.0", "coord_mlp_vel.0", "gravity_mlp.0", "node_mlp.0",
               "node_mlp.2", "node_mlp_virtual.0", "node_mlp_virtual.2", "embedding_in")


def _pad_layout(name: str, shape, h: int, C_: int, rf: bool):
    """The 64-wide image of a parameter of a ``hidden_nf = h < 64`` model: every hidden-sized row / column block is
    zero-extended to 64.  A zero row of a Linear gives a zero pre-activation, SiLU(0) = 0, and a zero column ignores
    its input, so the padded model computes exactly the reference's function (models/FastEGNN.py:28-99 with
    hidden_nf = h).  Returns (rows, cols, rows_dst, blocks, out_shape[, lead]): the parameter seen as a [rows, cols]
    matrix, the padded row count, the source-column blocks that are widened to 64 per h columns (the hidden-sized pieces
    of the reference's torch.cat inputs, FastEGNN.py:104,114,157,171; the remaining columns follow unchanged), the shape
    of the image and -- EGNN only -- the unchanged columns in front of the blocks.  tests/test_pad_cpu.py evaluates the
    oracle on images built from this layout."""
    shape = tuple(shape)
    if name == "virtual_node_feat":                       # [1, h, C]
        return h, shape[2], H, [], (1, H, shape[2])
    mod, kind = name.rsplit(".", 1)
    mod = mod.split(".", 1)[1] if mod.startswith("gcl_") else mod
    if kind == "bias":
        if mod in _HIDDEN_OUT:
            return 1, shape[0], 1, [h], (H,)
        return 1, shape[0], 1, [], shape
    if mod in ("edge_mlp.0", "edge_mlp_virtual.0", "node_mlp_virtual.0"):
        blocks = [h, h]
    elif mod == "node_mlp.0":
        blocks = [h, h, h * C_]                           # flat(v) is feature-major: column 2h + i*C + c
    elif mod == "embedding_in" or (rf and mod == "coord_mlp_vel.0"):
        blocks = []                                       # inputs that are not hidden features
    else:
        blocks = [h]
    rows, cols = shape
    rows_dst = H if mod in _HIDDEN_OUT else rows
    cols_dst = cols + sum(b // h * (H - h) for b in blocks)
    return rows, cols, rows_dst, blocks, (rows_dst, cols_dst)


def _activation_kind(act_fn: nn.Module):
    """(FASTEGNN_ACT_* kind, parameter) of the reference constructor's ``act_fn`` (models/FastEGNN.py:227)."""
    if isinstance(act_fn, nn.SiLU):
        return K.ACT_SILU, 0.0
    if isinstance(act_fn, nn.ReLU):
        return K.ACT_RELU, 0.0
    if isinstance(act_fn, nn.LeakyReLU):
        return K.ACT_LEAKY_RELU, float(act_fn.negative_slope)
    if isinstance(act_fn, nn.Tanh):
        return K.ACT_TANH, 0.0
    if isinstance(act_fn, nn.Sigmoid):
        return K.ACT_SIGMOID, 0.0
    if isinstance(act_fn, nn.ELU):
        return K.ACT_ELU, float(act_fn.alpha)
    if isinstance(act_fn, nn.GELU) and getattr(act_fn, "approximate", "none") == "none":
        return K.ACT_GELU, 0.0
    if isinstance(act_fn, nn.Softplus) and float(act_fn.threshold) == 20.0:
        return K.ACT_SOFTPLUS, float(act_fn.beta)
    raise NotImplementedError("fastegnn_amd: act_fn must be one of SiLU, ReLU, LeakyReLU, Tanh, Sigmoid, ELU, GELU "
                              f"(exact) or Softplus(threshold=20); got {act_fn!r}")


class _PadParams(torch.autograd.Function):
    """All parameters of a narrow model -> their 64-wide images, one HIP launch per 64 parameters
    (fastegnn_pad_params); the backward slices the padded gradients back with the same kernel in reverse."""

    @staticmethod
    def _table(layouts, narrow, wide):
        tab = (K.PadDesc * len(layouts))()
        for d, lay, a, b in zip(tab, layouts, narrow, wide):
            rows, cols, rows_dst, blocks = lay[:4]
            d.lead = lay[5] if len(lay) > 5 else 0
            d.src, d.dst = a.data_ptr(), b.data_ptr()
            d.rows, d.cols, d.rows_dst = rows, cols, rows_dst
            d.cols_dst = b.numel() // max(rows_dst, 1)
            d.nblk = len(blocks)
            for i, blk in enumerate(blocks):
                d.blk[i] = blk
        return tab

    @staticmethod
    def forward(ctx, names, h, C_, rf, *params):
        lib = K.lib()
        dev = params[0].device
        # rf: FastRF flag, or a layout function (name, shape, h) -> layout for another parameter naming (fastegnn_amd.egnn)
        layouts = [(rf(n, p.shape, h) if callable(rf) else _pad_layout(n, p.shape, h, C_, rf)) for n, p in zip(names, params)]
        sizes = [(math.prod(l[4]) + 3) // 4 * 4 for l in layouts]      # 16-byte aligned pieces of one buffer
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        outs, off = [], 0
        for l, sz in zip(layouts, sizes):
            outs.append(flat[off:off + math.prod(l[4])].view(l[4]))
            off += sz
        src = [p.detach().contiguous().float() for p in params]
        tab = _PadParams._table(layouts, src, outs)
        K.check(lib.fastegnn_pad_params(tab, len(layouts), h, 0, _stream(dev)), "fastegnn_pad_params")
        ctx.layouts, ctx.h, ctx.shapes = layouts, h, [p.shape for p in params]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        lib = K.lib()
        idx = [i for i, g in enumerate(gouts) if g is not None]
        grads = [None] * len(gouts)
        if idx:
            dev = gouts[idx[0]].device
            wide = [gouts[i].contiguous().float() for i in idx]
            sizes = [(math.prod(ctx.shapes[i]) + 3) // 4 * 4 for i in idx]
            flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
            narrow, off = [], 0
            for i, sz in zip(idx, sizes):
                narrow.append(flat[off:off + math.prod(ctx.shapes[i])].view(ctx.shapes[i]))
                off += sz
            tab = _PadParams._table([ctx.layouts[i] for i in idx], narrow, wide)
            K.check(lib.fastegnn_pad_params(tab, len(idx), ctx.h, 1, _stream(dev)), "fastegnn_pad_params")
            for i, g in zip(idx, narrow):
                grads[i] = g
        return (None, None, None, None, *grads)


class _FastEGNNFunction(torch.autograd.Function):
    """Whole-model forward/backward on the HIP library (one C-ABI call per layer and direction)."""

    @staticmethod
    def forward(ctx, spec: _Spec, graph: SortedGraph, batch32, gptr, edge_attr, node_attr,
                node_feat, node_loc, node_vel, loc_mean, *params):
        lib = K.lib(act=spec.act_kind != K.ACT_SILU, wide=spec.wide)
        ctx.wide = spec.wide   # the backward reads what this build saved (P / Q in units of ln 2 on the f16x2 build): same build
        ctx.guard = getattr(spec, "guard", None)   # RangeGuard of the module when this forward runs on the f16x2 build
        # edge_attr / node_attr are differentiable inputs (the reference harness detaches them, utils/train.py:33,46-47,
        # but the module itself is differentiable in them): their gradients are accumulated by the edge / virtual backward
        # kernels when asked for
        ea_sorted = graph.permute(edge_attr.detach() if edge_attr is not None else None)
        node_attr = node_attr.detach().contiguous().float() if node_attr is not None else None
        dev = node_loc.device
        st = _stream(dev)
        N, B, Cn = node_loc.size(0), loc_mean.size(0), spec.C
        f32 = dict(dtype=torch.float32, device=dev)
        params = [p.detach() for p in params]
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() % 16:
                raise RuntimeError("fastegnn_amd: parameters must be contiguous, 16-byte aligned fp32 tensors")
        node_feat, node_loc, node_vel, loc_mean = (t.detach().contiguous().float()
                                                   for t in (node_feat, node_loc, node_vel, loc_mean))
        h = torch.empty(N, H, **f32)
        K.check(lib.fastegnn_embed_forward(K.ptr(node_feat), N, spec.nf, K.ptr(params[1]), K.ptr(params[2]),
                                           K.ptr(h), st), "fastegnn_embed_forward")
        HvT = torch.empty(B, Cn, H, **f32)
        K.check(lib.fastegnn_virtual_init(K.ptr(params[0]), B, Cn, K.ptr(HvT), st), "fastegnn_virtual_init")
        x, Z = node_loc, loc_mean
        saved = []
        nwp = (lib.fastegnn_wpack_floats(Cn) + 3) // 4 * 4
        # the weight images of ALL layers in one launch (fastegnn_pack_weights_all): four launches of 34 + 2C small workgroups
        # were four launch latencies in front of the step
        wpack_all = torch.empty(spec.n_layers, nwp, **f32)
        tabs = [_PtrTable([params[s] if s is not None else None for s in spec.layer_slots[i]]) for i in range(spec.n_layers)]
        packs = (C.POINTER(K.LayerT) * spec.n_layers)()
        keep = []
        for i in range(spec.n_layers):
            Lp = _new_layer(spec, N, B, graph)
            _fill(Lp, params=tabs[i].addr(), wpack=wpack_all[i])
            keep.append(Lp)
            packs[i] = C.pointer(Lp)
        K.check(lib.fastegnn_pack_weights_all(packs, spec.n_layers, st), "fastegnn_pack_weights_all")
        for i in range(spec.n_layers):
            tab = tabs[i]
            b = dict(h=h, x=x, Z=Z, HvT=HvT, wpack=wpack_all[i])
            b.update(_carve(dev, dict(
                P=(N, H), QX=(N, K.QX_LD), A=(N, H), svel=(N,), sgrav=(N,), xsum=(B, 4),
                Bc=(B, Cn, H), aggm=(N, H), npre=(N, H), poolV=(B, Cn, H))))
            # outputs / forward-only scratch: separate allocations so that they can be freed individually
            b.update(_carve(dev, dict(aggx=(N, 3), poolX=(B, 3, Cn))))
            b.update(h_out=torch.empty(N, H, **f32), x_out=torch.empty(N, 3, **f32),
                     Z_out=torch.empty(B, 3, Cn, **f32), HvT_out=torch.empty(B, Cn, H, **f32))
            L = _new_layer(spec, N, B, graph)
            L.flags |= K.F_WPACK_READY
            _fill(L, batch=batch32, gptr=gptr, vel=node_vel, params=tab.addr(), **b)
            L.QX_src = b["QX"].data_ptr()
            if ea_sorted is not None:
                L.ea_sorted = ea_sorted.data_ptr()
            if node_attr is not None:
                L.node_attr = node_attr.data_ptr()
            K.check(lib.fastegnn_layer_forward(C.byref(L), st), f"fastegnn_layer_forward[{i}]")
            saved.append(b)
            h, x, Z, HvT = b["h_out"], b["x_out"], b["Z_out"], b["HvT_out"]
            # outputs of layer i are the inputs of layer i+1; drop what backward does not need
            for k in ("aggx", "poolX", "h_out", "x_out", "Z_out", "HvT_out"):
                del b[k]
        ctx.spec, ctx.graph, ctx.saved = spec, graph, saved
        ctx.misc = (batch32, gptr, ea_sorted, node_attr, node_feat, node_vel, params)
        return x, Z

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loc, g_vloc):
        lib = K.lib(act=ctx.spec.act_kind != K.ACT_SILU, wide=ctx.wide)
        spec, graph, saved = ctx.spec, ctx.graph, ctx.saved
        guard = ctx.guard
        if guard is not None:
            guard.poll("FastEGNN", ctx.misc[-1])   # (no synchronisation: if the forward's guard launch has run and tripped, switch now)
        if saved is None or any(b is None for b in saved):
            raise RuntimeError("fastegnn_amd: backward through the graph a second time: the saved stage products are "
                               "freed layer by layer during the first backward (retain_graph is not supported)")
        batch32, gptr, ea_sorted, node_attr, node_feat, node_vel, params = ctx.misc
        dev = node_vel.device
        st = _stream(dev)
        N, B, Cn, E = node_vel.size(0), saved[0]["Z"].size(0), spec.C, graph.E
        f32 = dict(dtype=torch.float32, device=dev)
        # gradient buffers: one flat zero-filled allocation, 16-byte aligned slices
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]
        flat = torch.zeros(sum(sizes), **f32)
        grads, off = [], 0
        for p, s in zip(params, sizes):
            grads.append(flat[off:off + p.numel()].view_as(p))
            off += s
        g_h = torch.zeros(N, H, **f32)
        g_x = (g_loc if g_loc is not None else torch.zeros(N, 3, **f32)).contiguous().float()
        g_Z = (g_vloc if g_vloc is not None else torch.zeros(B, 3, Cn, **f32)).contiguous().float()
        g_HvT = torch.zeros(B, Cn, H, **f32)
        g_vel = torch.zeros(N, 3, **f32)
        want_ea = ctx.needs_input_grad[4] and ea_sorted is not None and E > 0
        want_na = ctx.needs_input_grad[5] and node_attr is not None
        g_ea_sorted = torch.zeros(E, spec.ea, **f32) if want_ea else None
        g_na = torch.zeros(N, spec.na, **f32) if want_na else None
        scratch = _carve(dev, dict(
            g_poolV=(B, Cn, H), g_poolX=(B, 3, Cn), g_Bc=(B, Cn, H), g_Zp=(B, 3, Cn), g_xbar=(B, 4),
            g_A=(N, H), g_P=(N, H), g_aggm=(N, H), g_aggx=(N, 3), g_svel=(N,), g_sgrav=(N,),
            g_QXe=(max(E, 1) if spec.flags & K.F_DETERMINISTIC else 1, K.QX_LD), g_QX_src=(N, K.QX_LD), g_xrow=(N, 3),
            wg_edge=(lib.fastegnn_wg_edge_floats(E),), wg_virt=(lib.fastegnn_wg_virt_floats_for(N, Cn, spec.flags),),
            wg_node=(lib.fastegnn_wg_node_floats(N, B, Cn),),
            wg_slab=(lib.fastegnn_wg_slab_floats(),)))
        for i in reversed(range(spec.n_layers)):
            b = saved[i]
            ptab = _PtrTable([params[s] if s is not None else None for s in spec.layer_slots[i]])
            gtab = _PtrTable([grads[s] if s is not None else None for s in spec.layer_slots[i]])
            L = _new_layer(spec, N, B, graph)
            out = dict(g_h=torch.empty(N, H, **f32), g_x=torch.empty(N, 3, **f32),
                       g_Z=torch.empty(B, 3, Cn, **f32), g_HvT=torch.empty(B, Cn, H, **f32))
            _fill(L, batch=batch32, gptr=gptr, vel=node_vel, params=ptab.addr(), grads=gtab.addr(),
                  g_h_out=g_h, g_x_out=g_x, g_Z_out=g_Z, g_HvT_out=g_HvT, g_vel=g_vel, **b, **out, **scratch)
            L.QX_src = b["QX"].data_ptr()
            L.g_QX = scratch["g_QX_src"].data_ptr()
            if ea_sorted is not None:
                L.ea_sorted = ea_sorted.data_ptr()
            if node_attr is not None:
                L.node_attr = node_attr.data_ptr()
            if g_ea_sorted is not None:
                L.g_ea_sorted = g_ea_sorted.data_ptr()
            if g_na is not None:
                L.g_node_attr = g_na.data_ptr()
            K.check(lib.fastegnn_layer_backward(C.byref(L), st), f"fastegnn_layer_backward[{i}]")
            g_h, g_x, g_Z, g_HvT = out["g_h"], out["g_x"], out["g_Z"], out["g_HvT"]
            saved[i] = None
        K.check(lib.fastegnn_virtual_init_backward(K.ptr(g_HvT), B, Cn, K.ptr(grads[0]), st),
                "fastegnn_virtual_init_backward")
        g_nf = torch.empty_like(node_feat) if ctx.needs_input_grad[6] else None
        g_ea = None
        if g_ea_sorted is not None:   # back to the caller's edge order: sorted edge k is input edge perm[k]
            g_ea = torch.empty_like(g_ea_sorted)
            g_ea.index_copy_(0, graph.perm[:E].long(), g_ea_sorted)
        K.check(lib.fastegnn_embed_backward(K.ptr(node_feat), K.ptr(g_h), N, spec.nf, K.ptr(params[1]),
                                            K.ptr(grads[1]), K.ptr(grads[2]), K.ptr(g_nf), st),
                "fastegnn_embed_backward")
        # The last layer's node_mlp / node_mlp_virtual only feed h and Hv, which nothing reads after the last layer: the
        # reference's autograd leaves their .grad None (and torch.optim.Adam then skips them); the kernels wrote zeros.
        if guard is not None:   # an overflowed forward (outputs non-finite) hands zero parameter gradients on, whenever the host learns of it
            guard.zero_if_flagged(lib, flat)
        last = spec.n_layers - 1
        for s_, suffix in zip(spec.layer_slots[last], K.PARAM_SLOTS):
            if s_ is not None and suffix.startswith(("node_mlp.", "node_mlp_virtual.")) and not (spec.flags & K.F_RF):
                grads[s_] = None
        return (None, None, None, None, g_ea, g_na, g_nf, g_x, g_vel, g_Z, *grads)


# ------------------------------------------------------------------------------------------
# the module
# ------------------------------------------------------------------------------------------
class FastEGNN(nn.Module):
    """MI355X-native drop-in for the reference ``FastEGNN`` (models/FastEGNN.py:226-276)."""

    _layer_cls = E_GCL_vel   # parameter holder of one layer
    _extra_flags = 0         # FASTEGNN_F_* bits a sibling model adds (fastrf.py)

    def __init__(self, node_feat_nf, node_attr_nf, edge_attr_nf, hidden_nf, virtual_channels, device='cpu',
                 act_fn=nn.SiLU(), n_layers=4, residual=True, attention=False, normalize=False, tanh=False,
                 gravity=None, *, mlp_dtype=torch.float32):
        """Reference signature (models/FastEGNN.py:227-228) plus one keyword-only extension: ``mlp_dtype`` --
        ``torch.float32`` (default: fp32-grade products) or ``torch.bfloat16`` (BASELINE configs[2]: the operands of
        every 64-wide contraction are rounded to bf16, fp32 accumulate; parameters, coordinates and all reductions
        stay fp32 -- FASTEGNN_F_BF16 in include/fastegnn_hip.h)."""
        super().__init__()
        if mlp_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("fastegnn_amd: mlp_dtype must be torch.float32 or torch.bfloat16")
        self.mlp_dtype = mlp_dtype
        assert virtual_channels > 0, f'Channels of virtual node must greater than 0 (got {virtual_channels})'
        # hidden_nf < 64 runs zero-padded on the 64-wide tiles (_pad_param); 64 < hidden_nf <= 256 takes the unfused WIDE path
        # (fastegnn_amd/wide.py: the reference's op sequence on generic-width HIP operators; FastEGNN, fp32 only)
        # ... and so does a model beyond the fused kernels' argument ceilings (virtual_channels > 64, edge_attr_nf > 7, node_feat_nf > 8):
        # the reference has none (models/FastEGNN.py:227-263), and the wide path's operators are generic in every width
        self._wide = hidden_nf > H or virtual_channels > 64 or edge_attr_nf > 7 or node_feat_nf > 8
        if not 1 <= hidden_nf <= 256:
            raise NotImplementedError(f"fastegnn_amd: hidden_nf must be at most 256 in this build (got {hidden_nf})")
        if self._wide and mlp_dtype != torch.float32:
            raise NotImplementedError("fastegnn_amd: hidden_nf > 64, virtual_channels > 64, edge_attr_nf > 7 or node_feat_nf > 8 run on "
                                      "the unfused wide path, which is built for fp32 operands only")
        self._act = _activation_kind(act_fn)
        if self._act[0] in (K.ACT_SIGMOID, K.ACT_SOFTPLUS) and hidden_nf < H:
            raise NotImplementedError("fastegnn_amd: hidden_nf < 64 runs zero-padded, which needs act_fn(0) = 0")
        if self._act[0] != K.ACT_SILU and mlp_dtype != torch.float32:
            raise NotImplementedError("fastegnn_amd: the bf16 operand mode is built for SiLU only")
        self.hidden_nf = hidden_nf
        self.device = device
        self.n_layers = n_layers
        self.virtual_channels = virtual_channels
        self.node_feat_nf, self.node_attr_nf, self.edge_attr_nf = node_feat_nf, node_attr_nf, edge_attr_nf
        self.residual, self.attention, self.normalize, self.tanh = residual, attention, normalize, tanh
        self.gravity = list(gravity) if gravity is not None else None
        self.virtual_node_feat = nn.Parameter(data=torch.randn(size=(1, hidden_nf, virtual_channels)),
                                              requires_grad=True)
        self.embedding_in = nn.Linear(node_feat_nf, self.hidden_nf)
        for i in range(n_layers):
            self.add_module("gcl_%d" % i, self._layer_cls(hidden_nf, node_attr_nf, edge_attr_nf, virtual_channels, act_fn,
                                                          attention, tanh, gravity))
        self._graph_cache: Dict[tuple, SortedGraph] = {}
        self.cache_graphs = True
        self._range = RangeGuard()
        self._deterministic = K.deterministic_default()
        self._spec = None      # built lazily (after .to(device) / load_state_dict), parameters are fixed objects
        self._spec_det = None  # the same with FASTEGNN_F_DETERMINISTIC, for the small graphs that take it by default
        self._plist = None
        self.to(self.device)

    @property
    def _param_index(self):
        return dict(self.named_parameters())

    @property
    def deterministic(self) -> bool:
        """False: the backward scatters the col-side adjoint of the edge stage with fp32 atomics -- results vary at
        rounding level from run to run, like torch's scatter_add_ on a GPU.  True: per-edge rows + a CSC-ordered sum
        (FASTEGNN_F_DETERMINISTIC): reproducible, 272 bytes of scratch per edge and a second index sort per graph.
        Unless set (here or by FASTEGNN_DETERMINISTIC=0/1) the faster form for the operand mode is used, as measured on
        one box: fp32 operands -> atomics (cfg4 13.09 vs 13.39 ms per step, cfg3 shape 9.36 vs 9.95, cfg2 4.16 vs 4.42;
        the kernel's own arithmetic covers the atomic unit's ~7 G cache-line updates per second), bf16 operands -> store +
        reduce (cfg4 10.68 vs 11.00, cfg3 7.67 vs 8.33: the shorter kernel would wait for the atomics)."""
        if self._deterministic is None:
            return self.mlp_dtype == torch.bfloat16
        return self._deterministic

    @deterministic.setter
    def deterministic(self, value):
        self._deterministic = None if value is None else bool(value)
        self._spec = None
        self._spec_det = None
        self._graph_cache = {}

    # graphs of at most this many edges take the col-keyed (order-independent) sum unless told otherwise: on a graph of a few dozen
    # nodes the arrival order of the atomic scatter moves the input gradients between runs at the level of the parity floor, and the
    # reference's CPU scatter_add_ is run-to-run exact; the extra index sort and reduce launch cost nothing measurable there
    SMALL_GRAPH_EDGES = 512

    def deterministic_for(self, n_edges: int) -> bool:
        """the form of the edge backward a call with `n_edges` edges takes (see `deterministic`)"""
        if self._deterministic is None and n_edges <= self.SMALL_GRAPH_EDGES:
            return True
        return self.deterministic

    def sorted_graph(self, edge_index: torch.Tensor, n_nodes: int, deterministic: Optional[bool] = None) -> SortedGraph:
        """CSR of `edge_index`, cached (``self.cache_graphs``, 8 entries) under (data_ptr, size, torch's version counter,
        n_nodes).  Caveat: a writer that bypasses torch's version counter -- a raw-pointer kernel such as this library's
        own ``fastegnn_radius_graph_fill`` refilling a reused buffer -- is not seen; pass a fresh tensor, or set
        ``cache_graphs = False``, when edge lists are rewritten in place that way."""
        det = self.deterministic_for(edge_index.size(1)) if deterministic is None else deterministic
        key = (edge_index.data_ptr(), edge_index.size(1), edge_index._version, n_nodes, n_nodes, 0, det)
        g = self._graph_cache.get(key) if self.cache_graphs else None
        if g is None:
            g = SortedGraph(edge_index, n_nodes, csc=det)
            if self.cache_graphs:
                if len(self._graph_cache) >= 8:
                    self._graph_cache.pop(next(iter(self._graph_cache)))
                g._keepalive = edge_index  # the key holds a data_ptr: keep the tensor alive with it
                self._graph_cache[key] = g
        return g

    def forward(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None,
                node_attr=None):
        if not node_loc.is_cuda:
            raise RuntimeError("fastegnn_amd.FastEGNN runs on a gfx950 GPU only (no CPU fallback): move the model "
                               "and its inputs to 'cuda'")
        if (edge_attr.size(1) if edge_attr is not None else 0) != self.edge_attr_nf:
            raise ValueError("edge_attr width does not match edge_attr_nf")
        if (node_attr.size(1) if node_attr is not None else 0) != self.node_attr_nf:
            raise ValueError("node_attr width does not match node_attr_nf")
        if self._wide:
            from . import wide
            if isinstance(edge_index, SortedGraph):
                raise TypeError("fastegnn_amd: the wide path (hidden_nf > 64) takes edge_index as the reference does, an int64 [2, E] tensor")
            if _DEBUG_CHECKS:
                _check_indices(edge_index, data_batch, node_loc.size(0), loc_mean.size(0))
            if edge_attr is not None and edge_attr.size(1) == 0:
                edge_attr = None
            return wide.forward(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr, node_attr)
        dev = node_loc.device
        N, B = node_loc.size(0), loc_mean.size(0)   # B from loc_mean: no .item() sync (cf. :267)
        if self._spec is None:
            self._spec = _Spec(self)
            pidx = self._param_index
            self._plist = [pidx[n] for n in self._spec.names]
        graph = edge_index if isinstance(edge_index, SortedGraph) else self.sorted_graph(edge_index, N)
        spec = self._spec
        # a graph that came with its col-keyed index (small graphs by default, `deterministic` otherwise) takes the order-independent sum
        if graph.cscptr is not None and not spec.flags & K.F_DETERMINISTIC:
            if getattr(self, "_spec_det", None) is None:
                self._spec_det = _Spec(self, deterministic=True)
            spec = self._spec_det
        elif graph.cscptr is None and spec.flags & K.F_DETERMINISTIC:
            raise ValueError("fastegnn_amd: deterministic=True needs a SortedGraph built with csc=True")
        lib = K.lib()
        if data_batch.dtype != torch.int64:      # the C entry point reads int64 (what PyG collate emits)
            data_batch = data_batch.long()
        if _DEBUG_CHECKS:
            _check_indices(edge_index, data_batch, N, B)
        batch32 = torch.empty(N, dtype=torch.int32, device=dev)
        gptr = torch.empty(B + 1, dtype=torch.int32, device=dev)
        K.check(lib.fastegnn_build_batch(K.ptr(data_batch.contiguous()), N, B, K.ptr(batch32), K.ptr(gptr), _stream(dev)),
                "fastegnn_build_batch")
        if edge_attr is not None and edge_attr.size(1) == 0:
            edge_attr = None
        plist = self._plist
        if self.hidden_nf < H:
            rf = bool(spec.flags & K.F_RF)
            plist = list(_PadParams.apply(tuple(spec.names), self.hidden_nf, spec.C, rf, *plist))
        guard, who = self._range, type(self).__name__
        guarded = spec.flags & K.F_BF16 == 0       # (the bf16 operand mode has fp32's exponent range)
        if guarded and not guard.wide:
            guard.poll(who, self._plist)           # plain host loads of two words: an overflow of an EARLIER pass switches the build here
        spec.wide = guard.wide
        spec.guard = guard if guarded and not guard.wide else None

        def run():
            return _FastEGNNFunction.apply(spec, graph, batch32, gptr, edge_attr, node_attr, node_feat, node_loc, node_vel,
                                           loc_mean, *plist)
        out = run()
        if guarded and not guard.wide:
            guard.launch(K.lib(act=spec.act_kind != K.ACT_SILU, wide=False), (out[0], out[1]), (node_loc, node_vel))
            if guard.mode == "sync" and guard.sync_and_poll(dev, who, self._plist):
                spec.wide, spec.guard = True, None
                out = run()        # the same call on the wide-range build; what it returns is what fp32 gives
        if _DEBUG_CHECKS and not (bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[1]).all())):
            raise FloatingPointError("fastegnn_amd: non-finite outputs (on the wide-range build as well: the inputs or the "
                                     "weights themselves overflow fp32)" if guard.wide else "fastegnn_amd: non-finite outputs")
        return out
