"""One-process-per-GPU data parallelism for the FastEGNN hot path (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Graphs are the independent units of the path: rank r of W runs forward+backward on its own graphs
(`shard_units`) and the parameter gradients -- 0.55 M fp32 = 2.2 MB, far below one xGMI link's
bandwidth-delay product -- are summed in ONE flat bucket (`allreduce_gradients`).  There is no
collective on the data path.  Sharding a single large graph over ranks is the job of the stage-level
C entry points (include/fastegnn_hip.h).
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as
    exported by torch.distributed.run).  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        # a rank that never arrives must not hang the others for the default 10-30 minutes
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ.get("FASTEGNN_DIST_TIMEOUT", "300")))
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_units(n_units: int, world: int, rank: int) -> range:
    """Contiguous, balanced slice of `n_units` graphs for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_units, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = False) -> int:
    """Sum (or average) the .grad of `params` over the group with one flat all-reduce; parameters
    whose .grad is None contribute zeros (the last layer's unused heads).  Returns bytes reduced."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return 0
    dev, dt = params[0].device, params[0].dtype
    staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"   # transport without device support

    def _all_reduce(flat):
        if staged:
            hflat = flat.cpu()
            dist.all_reduce(hflat, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(hflat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    # Fast path: the backward of fastegnn_amd.FastEGNN hands autograd slices of ONE flat buffer, and autograd
    # keeps them as the .grad tensors -- reduce that buffer in place (no gather / scatter copies, one collective).
    # (parameters without a gradient -- the last layer's unused heads -- are the same on every rank and stay None)
    with_grad = [p for p in params if p.grad is not None]
    if with_grad and all(p.grad.dtype == dt and p.grad.is_contiguous() for p in with_grad):
        stor = with_grad[0].grad.untyped_storage()
        if all(p.grad.untyped_storage().data_ptr() == stor.data_ptr() for p in with_grad):
            flat = torch.empty(0, device=dev, dtype=dt).set_(stor, 0, (stor.nbytes() // with_grad[0].grad.element_size(),))
            _all_reduce(flat)
            if average:
                flat.div_(world)
            return flat.numel() * flat.element_size()
    sizes = [p.numel() for p in params]
    flat = torch.zeros(sum(sizes), device=dev, dtype=dt)
    off = 0
    for p, n in zip(params, sizes):
        if p.grad is not None:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        off += n
    _all_reduce(flat)
    if average:
        flat.div_(world)
    off = 0
    for p, n in zip(params, sizes):
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return flat.numel() * flat.element_size()


def max_over_ranks(value: float, device) -> float:
    if dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
