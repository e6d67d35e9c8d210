"""CPU, world_size 2 and 4 over gloo: the graph-sharded path (fastegnn_amd/sharded.py) -- row-owner
partition, exchange of the source table (halo: all-to-all-v of the ghost rows found by HaloPlan, or the
all-gather of the whole table), all-reduce of centroid sums / virtual-node pools and their adjoints, the
transposed exchange of the source-table gradient, rank-0-only gradients of the replicated per-graph stages,
optional Morton reordering of the nodes -- driven on CPU with the oracle's stage functions as the compute
backend (tests/cpu_stage_backend.py).  Uneven shards (23 nodes over 4 ranks: 6, 6, 6, 5) and a graph that
spans three ranks.  Every rank's outputs and the all-reduced parameter gradients must equal the
single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fastegnn_amd
from fastegnn_amd.dist import allreduce_gradients, init_from_env
from fastegnn_amd.sharded import ShardedFastEGNN, ShardPlan
from oracle import fastegnn_ref as R
from tests.cpu_stage_backend import CpuOracleBackend
from tests.helpers import rel_err

CFG = dict(node_feat_nf=2, node_attr_nf=0, edge_attr_nf=2, hidden_nf=64, virtual_channels=4, n_layers=2)
SIZES = [14, 4, 5]    # 23 nodes: W=2 (Npad 12): graph 0 straddles the boundary; W=4 (Npad 6): graph 0 spans ranks 0, 1, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    g = torch.Generator().manual_seed(11)
    rows, cols, batch, off = [], [], [], 0
    for b, n in enumerate(SIZES):
        e = 5 * n
        rows.append(torch.randint(0, n, (e,), generator=g) + off)
        cols.append(torch.randint(0, n, (e,), generator=g) + off)
        batch += [b] * n
        off += n
    ei = torch.stack([torch.cat(rows), torch.cat(cols)])
    ei = ei[:, torch.randperm(ei.size(1), generator=g)]
    batch = torch.tensor(batch)
    N, B, C = off, len(SIZES), CFG["virtual_channels"]
    loc = torch.randn(N, 3, generator=g)
    cm = torch.zeros(B, 3).index_add_(0, batch, loc) / torch.bincount(batch).unsqueeze(1)
    inp = dict(node_feat=torch.rand(N, 2, generator=g), node_loc=loc, node_vel=torch.randn(N, 3, generator=g) * 0.3,
               edge_index=ei, data_batch=batch, loc_mean=cm.unsqueeze(-1).repeat(1, 1, C),
               edge_attr=torch.rand(ei.size(1), 2, generator=g))
    target = loc + torch.randn(N, 3, generator=g) * 0.2
    return inp, target


def _model(gravity, hidden=64):
    torch.manual_seed(7)
    m = fastegnn_amd.FastEGNN(CFG["node_feat_nf"], 0, CFG["edge_attr_nf"], hidden, CFG["virtual_channels"],
                              n_layers=CFG["n_layers"], gravity=gravity)
    with torch.no_grad():
        for k, v in m.named_parameters():
            if k.endswith((".coord_mlp_r.2.weight", "coord_mlp_r_virtual.2.weight", "coord_mlp_v_virtual.2.weight")):
                v.mul_(100.0)
    return m


def _loss(loc_rows, vloc, target_rows, n_total):
    return ((loc_rows - target_rows) ** 2).sum() / (3 * n_total) + 0.1 * vloc.pow(2).mean()


def _worker(rank, world, port, gravity, exchange, reorder, q, hidden=64):
    # (FASTEGNN_SHARDED_SPLIT=1: the [interior | boundary] launch ranges, which are on by default only with the asynchronous
    # schedule, are what this test is to cover)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FASTEGNN_SHARDED_SPLIT="1")
    init_from_env("gloo")
    inp, target = _inputs()
    m = _model(gravity, hidden)
    cfg = R.Config(**CFG, gravity=gravity)      # the stages see the 64-wide (zero-padded) parameters
    spec_names = None
    be = CpuOracleBackend(spec_names, CFG["n_layers"], cfg)
    sm = ShardedFastEGNN(m, backend=be, exchange=exchange)
    loc, vloc = sm.forward_local(sm.shard_inputs(**inp, reorder=reorder))
    plan = sm.plan
    assert plan.mode == exchange
    _loss(loc, vloc, plan.rows(target), target.size(0)).backward()
    allreduce_gradients(m.parameters())
    grads = {k: (p.grad.numpy().copy() if p.grad is not None else None) for k, p in m.named_parameters()}
    ids = plan.node_ids.numpy().copy() if plan.node_ids is not None else np.arange(plan.n0, plan.n1)
    grads["__parts__"] = [(r0, n, bool(h)) for r0, n, h in plan.parts]
    q.put((rank, ids, plan.exchanged_bytes(), loc.detach().numpy().copy(), vloc.detach().numpy().copy(), grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,gravity,exchange,reorder,hidden", [
    (2, None, "halo", False, 64), (2, [0, -1, 0], "allgather", False, 64), (2, [0, -1, 0], "halo", True, 64),
    (4, [0, -1, 0], "halo", False, 64), (4, None, "halo", True, 64), (4, None, "allgather", True, 64),
    (2, [0, -1, 0], "halo", True, 24)])
def test_sharded_graph_matches_single_process_oracle(world, gravity, exchange, reorder, hidden):
    """(hidden 24: the ranks pad the parameters to the stages' 64-wide tiles -- the pad op of the backend -- and the padded
    gradients are sliced back before the all-reduce; the single-process oracle runs at the true width)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gravity, exchange, reorder, q, hidden)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    # single-process oracle
    inp, target = _inputs()
    m = _model(gravity, hidden)
    cfg = R.Config(**{**CFG, "hidden_nf": hidden}, gravity=gravity)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in m.named_parameters()}
    loc, vloc = R.forward(p, cfg, **inp)
    _loss(loc, vloc, target, target.size(0)).backward()
    seen = np.concatenate([r[1] for r in res])
    assert sorted(seen.tolist()) == list(range(target.size(0)))        # every node owned by exactly one rank
    last = CFG["n_layers"] - 1
    split_seen = False
    for rank, ids, xbytes, loc_r, vloc_r, grads in res:
        parts = grads.pop("__parts__")
        assert sum(n for _, n, _ in parts) == len(ids)
        split_seen |= [h for _, _, h in parts] == [False, True]
        assert rel_err(loc_r, loc.detach()[torch.from_numpy(ids)]) < 1e-5
        assert rel_err(vloc_r, vloc.detach()) < 1e-5
        if exchange == "halo":                                             # ghosts are at most the remote nodes
            assert xbytes <= (target.size(0) - len(ids)) * 68 * 4
        for k, v in p.items():
            if v.grad is None:      # the last layer's unused heads: None here too (torch.optim.Adam skips them)
                assert k.startswith(f"gcl_{last}.node_mlp") and (grads[k] is None or not np.any(grads[k])), k
                continue
            assert grads[k] is not None, k
            assert rel_err(grads[k], v.grad) < 2e-4, (rank, k, rel_err(grads[k], v.grad))
    if exchange == "halo" and world == 2:
        # rank 0's rows (12 of graph 0's 14 nodes) lie in one graph: it orders them [interior | boundary] and launches the edge
        # stage twice (with four ranks every row of a 6-node shard has a ghost column: one launch)
        assert split_seen


def test_morton_order_keeps_graphs_contiguous_and_localises_neighbours():
    from fastegnn_amd.sharded import morton_order
    g = torch.Generator().manual_seed(3)
    loc = torch.rand(4000, 3, generator=g)
    batch = torch.cat([torch.zeros(2500, dtype=torch.long), torch.ones(1500, dtype=torch.long)])
    order = morton_order(loc, batch)
    assert sorted(order.tolist()) == list(range(4000))
    assert torch.equal(batch[order], batch)                                # graphs stay contiguous and in place
    # index distance of spatial neighbours shrinks: a contiguous range is a compact region
    lo = loc[order][:2500]
    d = torch.cdist(lo[:300], lo)
    nn_idx = d.topk(8, largest=False).indices
    spread = (nn_idx - torch.arange(300).unsqueeze(1)).abs().float().median()
    assert spread < 400, spread                                            # random order: ~800 for 2500 nodes


def test_shard_plan():
    for n, w in ((23, 2), (100, 8), (8, 8), (10, 4)):
        plans = [ShardPlan(n, w, r) for r in range(w)]
        assert plans[0].n0 == 0 and plans[-1].n1 == n
        assert all(a.n1 == b.n0 for a, b in zip(plans[:-1], plans[1:]))
        assert all(p.n_src == w * plans[0].Npad >= n for p in plans)
    with pytest.raises(ValueError):
        ShardPlan(9, 4, 3)       # ceil(9/4)=3 rows per rank: the last rank would own nothing
