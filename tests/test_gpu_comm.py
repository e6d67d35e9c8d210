"""-m gpu: the exchange steps behind the C ABI (fastegnn_comm_*, csrc/comm.hip) and the multi-rank bench flow.

The pool's boxes have ONE GPU and RCCL refuses two ranks on one device, so what can run here is: RCCL bound through
dlopen and brought up as a single-rank communicator, every collective of the ABI on it (a one-rank all-gather /
reduce-scatter / all-to-all-v is a copy, a one-rank all-reduce the identity -- enough to pin the argument conventions
and the stream ordering), capture of a collective into a HIP graph next to a kernel, the halo pack / unpack kernels
against torch, and `bench.py --gpus 2` end to end with two ranks sharing the device over gloo.  RCCL with more than
one rank has never run on this pool: DESIGN.md says so."""
import json
import os
import subprocess
import sys

import pytest
import torch

from fastegnn_amd import _lib as K

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm():
    from fastegnn_amd.comm import AbiComm
    c = AbiComm("cuda:0")
    yield c
    c.close()


def test_single_rank_collectives_through_the_abi(comm):
    assert comm.world == 1 and comm.rank == 0
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 68, generator=g).cuda()
    y = x.clone()
    comm.all_reduce(y).wait()
    out = torch.empty(5, 68, device="cuda")
    comm.all_gather(out, x).wait()
    rs = torch.empty(5, 68, device="cuda")
    comm.reduce_scatter(rs, x).wait()
    a2a = torch.zeros(5, 68, device="cuda")
    comm.all_to_all_v(a2a, x, [5], [5]).wait()
    empty = torch.zeros(0, 68, device="cuda")
    comm.all_to_all_v(empty, empty, [0], [0]).wait()
    torch.cuda.synchronize()
    for t in (y, out, rs, a2a):
        assert torch.equal(t, x)


def test_collective_is_capturable_into_a_hip_graph(comm):
    """The point of the ABI transport: a collective enqueued on the capture stream becomes a node of the graph, ordered
    with the kernels around it (torch.distributed runs its collectives on a stream of its own)."""
    x = torch.ones(4096, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.mul_(1.0)
        comm.all_reduce(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        x.mul_(2.0)              # kernel -> collective -> kernel
        comm.all_reduce(x)
        x.add_(1.0)
    x.fill_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, torch.full_like(x, 15.0))      # ((1*2+1)*2+1)*2+1


def test_halo_pack_and_unpack_kernels(comm):
    g = torch.Generator().manual_seed(1)
    table = torch.randn(300, 68, generator=g).cuda()
    ids = torch.randint(0, 300, (1000,), generator=g).cuda()
    rows = comm.gather_rows(table, ids)
    assert torch.equal(rows, table[ids])
    upd = torch.randint(-8, 8, (1000, 68), generator=g).float().cuda()      # integers: the atomic sums are exact
    t2 = torch.zeros(300, 68, device="cuda")
    comm.scatter_add_rows(t2, ids, upd)
    assert torch.equal(t2, torch.zeros(300, 68, device="cuda").index_add_(0, ids, upd))


def test_bench_two_ranks_over_gloo_prints_one_json_line():
    """`python bench.py --gpus 2` as the driver launches it for N > 1, with the two ranks sharing this box's one GPU over
    gloo (FASTEGNN_BENCH_BACKEND=gloo): fresh child processes, the partitioned frame with the Morton-ordered halo
    exchange, the data-parallel leg, ONE JSON line on stdout."""
    env = dict(os.environ, FASTEGNN_BENCH_BACKEND="gloo", FASTEGNN_BENCH_TIMEOUT="600")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--nodes", "20000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["nodes"] == 20000 and "sharding" in d["config"]["parallelism"]
    assert d["table_exchange"]["mode"] == "halo"
    assert 0 < d["table_exchange"]["bytes_received_per_exchange"] < 0.3 * d["table_exchange"]["all_gather_bytes_for_comparison"]
    assert {"xsum", "pools", "g_pools", "QX_halo", "g_QX_halo"} <= set(d["collectives"])
    leg = d["data_parallel_leg"]
    assert leg["scaling"] == "weak" and leg["value"] > 0
