"""RCCL collectives through the C ABI (include/fastegnn_hip.h, csrc/comm.hip): the transport of the sharded path
when FASTEGNN_COMM=abi.  Every call is enqueued on the current torch stream -- ordered with the stage kernels that
were launched before it and with those launched after it, so the handles returned here have nothing to wait for -- and
can be captured into a HIP graph together with them (the torch.distributed transport runs its collectives on a stream
of its own and cannot).  A non-torch host gets the same transport from the same entry points.

Bring-up uses torch.distributed only to hand the 128-byte RCCL unique id from rank 0 to the other ranks (any
backend; a single-process world needs no process group at all)."""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
import torch.distributed as dist

from . import _lib as K


class _Ordered:
    """Handle of a stream-ordered collective: its consumers are ordered behind it by the stream itself."""

    def wait(self):
        return True


class AbiComm:
    def __init__(self, device, group=None):
        self.lib = K.lib()
        self.dev = torch.device(device)
        self.group = group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        n = self.lib.fastegnn_comm_unique_id_bytes()
        uid = (C.c_char * n)()
        if rank == 0:
            K.check(self.lib.fastegnn_comm_unique_id(uid), "fastegnn_comm_unique_id")
        if world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = (C.c_char * n).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        with torch.cuda.device(self.dev):
            K.check(self.lib.fastegnn_comm_init(C.byref(self.handle), uid, rank, world), "fastegnn_comm_init")
        self.world, self.rank = world, rank

    def close(self):
        if self.handle:
            self.lib.fastegnn_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _st(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def _f32(t):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("AbiComm: contiguous fp32 CUDA tensors only")
        return t

    def all_reduce(self, t):
        K.check(self.lib.fastegnn_comm_all_reduce(self.handle, K.ptr(self._f32(t)), t.numel(), self._st()), "fastegnn_comm_all_reduce")
        return _Ordered()

    def all_gather(self, out, inp):
        assert out.numel() == self.world * inp.numel()
        K.check(self.lib.fastegnn_comm_all_gather(self.handle, K.ptr(self._f32(inp)), K.ptr(self._f32(out)), inp.numel(), self._st()),
                "fastegnn_comm_all_gather")
        return _Ordered()

    def reduce_scatter(self, out, inp):
        assert inp.numel() == self.world * out.numel()
        K.check(self.lib.fastegnn_comm_reduce_scatter(self.handle, K.ptr(self._f32(inp)), K.ptr(self._f32(out)), out.numel(), self._st()),
                "fastegnn_comm_reduce_scatter")
        return _Ordered()

    def all_to_all_v(self, out, inp, out_rows: List[int], in_rows: List[int]):
        """Rows of the trailing dimension's width: in_rows[r] rows of `inp` go to rank r, out_rows[r] rows arrive from it."""
        w = inp.size(-1) if inp.dim() > 1 else 1
        assert len(out_rows) == self.world and len(in_rows) == self.world
        assert sum(out_rows) * w == out.numel() and sum(in_rows) * w == inp.numel()
        sr = (C.c_int64 * self.world)(*in_rows)
        rr = (C.c_int64 * self.world)(*out_rows)
        K.check(self.lib.fastegnn_comm_all_to_all_v(self.handle, K.ptr(self._f32(inp)) if inp.numel() else None, sr,
                                                    K.ptr(self._f32(out)) if out.numel() else None, rr, w, self._st()),
                "fastegnn_comm_all_to_all_v")
        return _Ordered()

    def gather_rows(self, table, ids):
        out = torch.empty(ids.numel(), table.size(1), dtype=torch.float32, device=table.device)
        K.check(self.lib.fastegnn_gather_rows(K.ptr(table), K.ptr(ids), ids.numel(), table.size(1), K.ptr(out), self._st()),
                "fastegnn_gather_rows")
        return out

    def scatter_add_rows(self, table, ids, rows):
        K.check(self.lib.fastegnn_scatter_add_rows(K.ptr(table), K.ptr(ids), ids.numel(), table.size(1), K.ptr(rows), self._st()),
                "fastegnn_scatter_add_rows")
