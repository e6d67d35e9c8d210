"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- unit sharding, the flat gradient
all-reduce (fastegnn_amd/dist.py) and the max-over-ranks timing -- with the oracle standing in
for the per-rank compute (the HIP kernels need a GPU): two ranks on two different frames must
end up with the gradients of the two-frame batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastegnn_amd.dist import allreduce_gradients, init_from_env, max_over_ranks, shard_units
from oracle import fastegnn_ref as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame(seed, C):
    g = torch.Generator().manual_seed(seed)
    N, E = 12, 40
    loc = torch.randn(N, 3, generator=g)
    return dict(node_feat=torch.rand(N, 2, generator=g), node_loc=loc, node_vel=torch.randn(N, 3, generator=g),
                edge_index=torch.randint(0, N, (2, E), generator=g), data_batch=torch.zeros(N, dtype=torch.long),
                loc_mean=loc.mean(0).view(1, 3, 1).repeat(1, 1, C), edge_attr=torch.rand(E, 2, generator=g))


def _grads(cfg, p, frame):
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loc, vloc = R.forward(pp, cfg, **frame)
    (loc.pow(2).mean() + vloc.pow(2).mean()).backward()
    return pp


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg = R.Config(2, 0, 2, 64, 3, n_layers=2)
    p = R.init_params(cfg, seed=5, coord_gain=0.1)
    mine = list(shard_units(2, world, rank))
    assert mine == [rank]
    pp = _grads(cfg, p, _frame(100 + mine[0], 3))
    params = [torch.nn.Parameter(v.detach()) for v in pp.values()]
    for prm, v in zip(params, pp.values()):
        prm.grad = None if v.grad is None else v.grad.clone()   # None for the last layer's unused heads
    nbytes = allreduce_gradients(params)
    t = max_over_ranks(1.0 + rank, "cpu")
    q.put((rank, [None if prm.grad is None else prm.grad.numpy().copy() for prm in params], nbytes, t))
    # the layout the HIP backward produces: every .grad a slice of ONE flat buffer (16-byte aligned slots)
    sizes = [(v.numel() + 3) // 4 * 4 for v in pp.values()]
    flat = torch.zeros(sum(sizes))
    off = 0
    for prm, v, n in zip(params, pp.values(), sizes):
        prm.grad = flat[off:off + v.numel()].view_as(v)
        if v.grad is not None:
            prm.grad.copy_(v.grad)
        off += n
    before = flat.data_ptr()
    nb2 = allreduce_gradients(params)
    assert nb2 == flat.numel() * 4 and all(prm.grad.untyped_storage().data_ptr() == before for prm in params)
    q.put((rank + 100, [prm.grad.numpy().copy() for prm in params], nb2, 0.0))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_two_frame_sum():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    out = dict()
    for _ in range(2 * world):
        rank, grads, nbytes, t = q.get(timeout=120)
        out[rank] = (grads, nbytes, t)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    cfg = R.Config(2, 0, 2, 64, 3, n_layers=2)
    p = R.init_params(cfg, seed=5, coord_gain=0.1)
    ref = [_grads(cfg, p, _frame(100 + r, 3)) for r in range(world)]
    keys = list(p.keys())
    for rank in range(world):
        grads, nbytes, t = out[rank]
        assert nbytes == sum(v.numel() for v in p.values()) * 4
        assert t == 2.0                                     # max over ranks of (1 + rank)
        for k, g in zip(keys, grads):
            want = sum((ref[r][k].grad if ref[r][k].grad is not None else torch.zeros_like(p[k])) for r in range(world))
            assert g is not None and torch.allclose(torch.from_numpy(g), want, rtol=1e-6, atol=1e-7), k
            g2 = out[rank + 100][0][keys.index(k)]   # in-place reduction of the shared flat buffer
            assert torch.allclose(torch.from_numpy(g2), want, rtol=1e-6, atol=1e-7), k


def test_shard_units_partitions():
    for n in (0, 1, 7, 8, 100):
        for w in (1, 2, 3, 8):
            parts = [list(shard_units(n, w, r)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(len(x) for x in parts) - min(len(x) for x in parts) <= 1
