#!/usr/bin/env python3
"""bench.py -- fwd+bwd throughput of the HIP-backed FastEGNN on the Water-3D-like 100k-node frame
(BASELINE.json configs[3] / SURVEY.md section 8d cfg4), one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one forward + loss + backward of the 4-layer model on one synthetic frame that is already
resident in HBM (COO edge_index as the reference receives it; the CSR build is inside the step).
With N > 1 every rank processes its own frame (graphs are the independent units of the metric ->
weak scaling) and the parameter gradients are all-reduced over RCCL inside the step, as a
data-parallel trainer would.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak
H = 64
UNIT = 2 * H * H               # FLOPs of one 64x64 mat-vec


def make_frame(N, C, seed, device, radius=0.035):
    """N points uniform in a box of the density of [0,0.965]^3 @ 100k (mean degree
    ~20 at r=0.035), vel ~ N(0,0.003^2), node_feat=[|vel|,1], edge_attr=[dist,dist], target=loc+20 vel."""
    g = torch.Generator().manual_seed(seed)
    box = 0.965 * (N / 100000.0) ** (1.0 / 3.0)
    loc = torch.rand(N, 3, generator=g) * box
    vel = torch.randn(N, 3, generator=g) * 0.003
    if str(device) != "cpu":
        # radius graph + length ordering on the GPU (fastegnn_amd/graphs.py; identical edge set, tests/test_gpu_graphs.py)
        from fastegnn_amd.graphs import cutoff_edges, radius_graph
        ei, dist = radius_graph(loc.to(device), radius)
        ei, dist = cutoff_edges(ei, dist, 0.0)                          # datasets emit edges sorted by length
        ei, dist = ei.cpu(), dist.cpu()
    else:
        from scipy.spatial import cKDTree
        pairs = cKDTree(loc.numpy().astype(np.float64)).query_pairs(radius, output_type="ndarray")
        pairs = torch.from_numpy(pairs.astype(np.int64))
        ei = torch.cat([pairs.t(), pairs.t().flip(0)], dim=1)              # both directions
        # datasets emit edges sorted by length (datasets/simulation/dataset.py:96-101)
        dist = (loc[ei[0]] - loc[ei[1]]).norm(dim=1)
        order = torch.argsort(dist)
        ei, dist = ei[:, order].contiguous(), dist[order]
    frame = dict(
        node_feat=torch.stack([vel.norm(dim=1), torch.ones(N)], 1),
        node_loc=loc, node_vel=vel, edge_index=ei,
        data_batch=torch.zeros(N, dtype=torch.long),
        loc_mean=loc.mean(0).view(1, 3, 1).repeat(1, 1, C),
        edge_attr=torch.stack([dist, dist], 1),
    )
    target = loc + 20.0 * vel
    return {k: v.to(device) for k, v in frame.items()}, target.to(device)


def loss_fn(loc, vloc, target):
    return torch.nn.functional.mse_loss(loc, target) + 0.01 * vloc.pow(2).mean()


def kernel_model(N, E, B, C, L, gravity=True):
    """Algorithmic (FLOPs, bytes) of each kernel summed over ONE step (L layers, fwd+bwd).
    FLOPs count the 64x64 contractions actually required (no tile padding); bytes count every
    operand once (DESIGN.md 'Roofline accounting')."""
    NC = N * C
    heads = 2 if gravity else 1
    f = {}
    f["edge_fwd_kernel"] = (L * E * 2 * UNIT, L * (280 * E + 540 * N))
    # edge backward: 2 recomputed + 2 transposed layers + the two in-workgroup weight-gradient contractions per edge
    f["edge_bwd_kernel"] = (L * E * 6 * UNIT, L * (E * (8 + 8 + 272 + 12 + 272 + 32) + N * (256 * 2 + 12 * 2 + 268)))
    f["virt_fwd_kernel"] = (L * (NC * 4 + N * 3) * UNIT, L * N * (5 * 256 + 60))
    f["virt_bwd_kernel"] = (L * (NC * 7 + N * 3) * UNIT, L * (NC * 5 * 256 + N * (8 * 256 + 60)))
    f["node_pre_fwd_kernel"] = (L * N * (3 + heads) * UNIT, L * N * (256 + 12 + 256 + 272 + 256 + 8))
    f["node_pre_bwd_kernel"] = (L * N * (3 + 2 * heads) * UNIT, L * N * (256 * 6 + 272 + 48))
    # wgrad launches per layer: virt (3 x NC + C x N for the per-channel node_mlp block),
    # node-level (N x (3 + 3 + heads)), graph-level (B*C x 5); the edge stage contracts in edge_bwd
    m_rows = L * (3 * NC + C * N + (6 + heads) * N + 5 * B * C)
    f["wgrad_tn_kernel"] = (m_rows * UNIT, m_rows * 512)
    f["wgrad_small_kernel"] = (L * E * 2 * 64 * 3, L * E * (272 + 32))
    f["edge_col_reduce_kernel"] = (0, L * (E * (272 + 4) + N * 272))
    return f


CPU_THREADS = 8   # fastest of {8,16,32,64,128} torch threads for this op mix on the GPU box's host CPU


def cpu_baseline(C, seed, steps=2, n_sample=10000):
    """Oracle (op-for-op CPU restatement of the reference) timed on the host cores on a bounded
    sample: a frame of the same density with n_sample nodes; cost is linear in N and E at fixed C."""
    from oracle import fastegnn_ref as R
    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or CPU_THREADS))
    frame, target = make_frame(n_sample, C, seed, "cpu")
    cfg = R.Config(node_feat_nf=2, node_attr_nf=0, edge_attr_nf=2, hidden_nf=64, virtual_channels=C,
                   n_layers=4, gravity=[0, -1, 0])
    p = {k: v.requires_grad_(True) for k, v in R.init_params(cfg, seed=43).items()}
    ts = []
    for i in range(steps + 1):
        for v in p.values():
            v.grad = None
        t0 = time.perf_counter()
        loc, vloc = R.forward(p, cfg, **frame)
        loss_fn(loc, vloc, target).backward()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[1:]))
    return t, frame["edge_index"].size(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=100000)
    ap.add_argument("--channels", type=int, default=16)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cache-graph", action="store_true", help="reuse the sorted graph across steps")
    ap.add_argument("--sharded", action="store_true",
                    help="ONE frame evaluated cooperatively by all ranks (row-owner sharding, fastegnn_amd/sharded.py: "
                         "all-gather of the source table, reduce-scatter of its gradient, tiny all-reduces) instead of "
                         "one frame per rank; strong scaling")
    ap.add_argument("--train-step", action="store_true",
                    help="time a full training iteration instead (edge_attr augmentation, MSE+MMD loss, Adam: "
                         "fastegnn_amd.train.train_step); the default step is fwd+loss+bwd, the BASELINE metric")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    import fastegnn_amd
    from fastegnn_amd import _lib as K
    from fastegnn_amd.dist import allreduce_gradients, init_from_env, max_over_ranks
    init_from_env("nccl")
    if args.sharded and not dist.is_initialized():   # the sharded path talks to a process group even at world 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    N, C, L = args.nodes, args.channels, args.layers
    torch.manual_seed(43)
    model = fastegnn_amd.FastEGNN(node_feat_nf=2, node_attr_nf=0, edge_attr_nf=2, hidden_nf=64, virtual_channels=C,
                                  device=dev, n_layers=L, gravity=[0, -1, 0])
    model.cache_graphs = bool(args.cache_graph)
    frame, target = make_frame(N, C, 43 + (0 if args.sharded else rank), dev)   # sharded: every rank holds the same frame
    E = frame["edge_index"].size(1)
    params = [p for p in model.parameters()]

    if args.train_step:
        from fastegnn_amd.train import FusedAdam, train_step
        opt = FusedAdam(params, lr=5e-4, weight_decay=1e-12)
        gsel = torch.Generator().manual_seed(7)
        samp = torch.randperm(N, generator=gsel)[: 3 * C].view(1, -1).to(dev)
        data = dict(loc_0=frame["node_loc"], vel_0=frame["node_vel"], loc_t=target, node_feat=frame["node_feat"],
                    edge_index=frame["edge_index"], edge_attr=frame["edge_attr"][:, :1].contiguous(),
                    batch=frame["data_batch"], loc_mean=frame["loc_mean"])

    if args.sharded:
        from fastegnn_amd.sharded import ShardedFastEGNN
        smodel = ShardedFastEGNN(model)

    def step():
        if args.train_step:   # utils/train.py:30-170 on device (single-GPU only)
            return train_step(model, opt, data, samp, 1.0, 0.01)[0]
        for p in params:
            p.grad = None
        if args.sharded:      # this rank's rows of the MSE + its 1/world share of the replicated virtual-node term
            loc, vloc = smodel(**frame)
            tgt = smodel.plan.rows(target)
            loss = (loc - tgt).pow(2).sum() / (3 * N) + 0.01 * vloc.pow(2).mean() / world
            loss.backward()
            allreduce_gradients(params)
            return loss
        loc, vloc = model(**frame)
        loss = loss_fn(loc, vloc, target)
        loss.backward()
        if world > 1:   # data-parallel gradient exchange (one flat bucket, 2.2 MB, RCCL over xGMI)
            allreduce_gradients(params)
        return loss

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the per-kernel HIP-event profiler is on during warm-up too, so that its event pool exists
    # before the timed region (hipEventCreate is slow on a cold driver)
    K.lib().fastegnn_profile_enable(1)
    # two extra untimed steps ahead of the W warm-up steps: on a freshly booted box the first pass still pages
    # the Python / torch / HIP code in from the image, which showed up as sporadic 100 ms host stalls
    for _ in range(2 + max(args.warmup, 1)):
        step()
    sync()
    K.profile_collect()
    t0 = time.perf_counter()
    cpu_ms = []
    done = []   # bound the CPU run-ahead to two steps (deep HIP queues stall sporadically on this stack)
    for _ in range(args.steps):
        tc = time.perf_counter()
        if len(done) >= 2:
            done.pop(0).synchronize()
        loss = step()
        ev = torch.cuda.Event()
        ev.record()
        done.append(ev)
        cpu_ms.append(1e3 * (time.perf_counter() - tc))
    sync()
    dt = time.perf_counter() - t0
    if os.environ.get("FASTEGNN_BENCH_DEBUG"):
        print("cpu enqueue ms per step:", " ".join(f"{t:.1f}" for t in cpu_ms), file=sys.stderr)
    K.lib().fastegnn_profile_enable(0)
    prof = K.profile_collect()
    dt = max_over_ranks(dt, dev)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = (1 if args.sharded else world) * args.steps / dt
        km = kernel_model(N, E, 1, C, L)
        kernels = {}
        for name, (ms, cnt) in prof.items():
            per_step = ms / args.steps
            ent = {"ms_per_step": round(per_step, 4), "launches_per_step": cnt / args.steps,
                   "avg_launch_ms": round(ms / cnt, 5)}
            if name in km:
                fl, by = km[name]
                ent["tflops"] = round(fl / (per_step * 1e-3) / 1e12, 3)
                ent["gbs"] = round(by / (per_step * 1e-3) / 1e9, 1)
            kernels[name] = ent
        dom = max((n for n in kernels if n in km), key=lambda n: kernels[n]["ms_per_step"])
        fl, by = km[dom]
        t_mfma, t_hbm = fl / (PEAK_MFMA_F32_TFLOPS * 1e12), by / (PEAK_HBM_GBS * 1e9)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):   # HBM bytes per launch from rocprofv3 PMC passes (tools/gpu_traffic.sh)
            traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
        if t_mfma >= t_hbm:
            roof = {"kernel": dom, "bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": PEAK_MFMA_F32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(kernels[dom]["tflops"] / PEAK_MFMA_F32_TFLOPS, 4),
                    "traffic": traffic}
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["gbs"], "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(kernels[dom]["gbs"] / PEAK_HBM_GBS, 4), "traffic": traffic}
        es = kernels.get("edge_fwd_kernel")
        edge_scatter = None
        if es:
            edge_scatter = {"kernel": "edge_fwd_kernel", "algorithmic_bytes_per_launch": 280 * E + 540 * N,
                            "achieved_GBs": es["gbs"], "frac_of_hbm_peak": round(es["gbs"] / PEAK_HBM_GBS, 4),
                            "achieved_TFLOPs": es["tflops"],
                            "frac_of_mfma_f32_peak": round(es["tflops"] / PEAK_MFMA_F32_TFLOPS, 4)}
        out = {
            "metric": "graphs/sec (fwd+bwd), Water-3D-like 100k-node frame", "value": round(value, 4),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg4 Water-3D-like frame (SURVEY 8d): uniform points, radius graph r=0.035, "
                                   "4-layer FastEGNN H=64, gravity on, "
                                   + ("full training iteration (augment+fwd+MSE/MMD+bwd+Adam)" if args.train_step else "fwd+loss+bwd")
                                   + ", CSR build "
                                   + ("cached" if args.cache_graph else "inside the step"),
                       "nodes": N, "edges": E, "virtual_channels": C, "layers": L, "graphs_per_step_per_gpu": 1,
                       "parallelism": (f"row-owner sharding of one frame over {world} GPU(s) (fastegnn_amd/sharded.py)" if args.sharded
                                       else f"dp{world} (one frame per GPU, RCCL gradient all-reduce)" if world > 1 else "1 GPU"),
                       "loss": float(loss.detach())},
            "roofline": roof,
            "edge_scatter": edge_scatter,
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:
            ns = 10000
            t_cpu, e_cpu = cpu_baseline(C, 43, steps=3, n_sample=ns)
            scale = N / ns
            out["cpu_baseline"] = {
                "value": round(1.0 / (t_cpu * scale), 5), "unit": "graphs/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"oracle/fastegnn_ref.py (torch CPU, op-for-op) fwd+bwd on a {ns}-node frame of the same "
                          f"density (E={e_cpu}), {torch.get_num_threads()} torch threads (fastest setting measured on this host), "
                          f"median of 3 steps = {t_cpu:.2f} s, scaled x{scale:.0f} to the "
                          f"{N}-node frame (cost is linear in N and E at fixed C)"}
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)   # flushed before any communicator teardown
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
