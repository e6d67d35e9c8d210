#!/usr/bin/env python3
"""bench.py -- fwd+bwd throughput of the HIP-backed FastEGNN on the BASELINE.json configurations, one
process per GPU.

    python bench.py                               # cfg4 (the headline 100k-node Water-3D-like frame), 1 GPU
    python bench.py --gpus 8                      # spawns its 8 ranks itself (one per GPU, RCCL over xGMI)
    python bench.py --config cfg3|cfg1|cfg2|cfg5  # the other BASELINE configurations
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # an external launcher works as well

A step = one forward + loss + backward of the 4-layer model on one synthetic batch that is already
resident in HBM (COO edge_index as the reference receives it; the CSR build is inside the step).

N > 1:
  cfg4 / cfg5 (one large graph): the frame is partitioned over the ranks (north_star's partition: row-owner
      sharding, fastegnn_amd/sharded.py -- all-gather of the source table, reduce-scatter of its gradient, tiny
      all-reduces of the virtual-node accumulators): STRONG scaling of one frame, `value` = frames/s of the job.
  cfg1..cfg3 (mini-batches of small graphs): graphs are the independent units; every rank runs its own
      mini-batch and the parameter gradients are all-reduced (`--mode dp`; also selectable for cfg4): WEAK.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA peak (= fp32 vector peak)
PEAK_MFMA_BF16_TFLOPS = 2500.0 # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak
H = 64
UNIT = 2 * H * H               # FLOPs of one 64x64 mat-vec

# BASELINE.json configs[0..4] as synthetic workloads
CONFIGS = {
    "cfg1": dict(kind="nbody", graphs=100, nodes=5, C=3, cutoff=0.5, gravity=None, dtype="f32",
                 text="cfg1 N-body: 100 graphs x 5 particles, 10 shortest directed pairs per graph (cutoff_rate 0.5), C=3"),
    "cfg2": dict(kind="nbody", graphs=100, nodes=100, C=3, cutoff=0.0, gravity=None, dtype="f32",
                 text="cfg2 N-body: 100 graphs x 100 particles, fully connected (990000 edges), C=3"),
    "cfg3": dict(kind="protein", graphs=8, nodes=3341, C=8, cutoff=0.5, gravity=None, dtype="bf16",
                 text="cfg3 protein-MD-like: 8 graphs x 3341 points uniform in a 36 A cube (+50 A offset), 10 A contacts "
                      "minus the longest 50 %, C=8, bf16 MLP operands / fp32 accumulate"),
    "cfg4": dict(kind="water", nodes=100000, C=16, gravity=[0, -1, 0], dtype="f32",
                 text="cfg4 Water-3D-like frame (SURVEY 8d): uniform points, radius graph r=0.035 (mean degree ~19), gravity on"),
    "cfg5": dict(kind="water", nodes=1000000, C=32, gravity=[0, -1, 0], dtype="f32",
                 text="cfg5 synthetic random geometric graph: 1 M points, r=0.035 at the cfg4 density (~19.6 M directed edges), gravity on"),
}


# ------------------------------------------------------------------------------------------------------
# launcher (parent process: never touches the GPU)
# ------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent
    has made no HIP call), relay rank 0's stdout (the JSON line) and return the worst exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Poll all ranks: if one dies (before or inside the rendezvous or a collective) the survivors would wait for it
    # for ever -- the first non-zero exit ends the others and becomes the exit code.  FASTEGNN_BENCH_TIMEOUT bounds the
    # whole run (default 30 min).
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("FASTEGNN_BENCH_TIMEOUT", "1800"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = abs(bad[0]) if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            sys.stderr.write(f"bench: rank exit codes {codes}: " + ("a rank failed" if bad else "timeout") + ", stopped the others\n")
            break
        if all(c is not None for c in codes):
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out = buf[0] if buf else b""
    # stdout carries the ONE JSON line; anything else rank 0 printed there (communicator banners) goes to stderr
    for line in out.decode().splitlines():
        if line.startswith('{"metric"'):
            sys.stdout.write(line + "\n")
        elif line.strip():
            sys.stderr.write(line + "\n")
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------------
# synthetic workloads
# ------------------------------------------------------------------------------------------------------
def make_frame(N, C, seed, device, radius=0.035):
    """N points uniform in a box of the density of [0,0.965]^3 @ 100k (mean degree
    ~19 at r=0.035), vel ~ N(0,0.003^2), node_feat=[|vel|,1], edge_attr=[dist,dist], target=loc+20 vel."""
    import numpy as np
    import torch
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


def make_nbody_batch(n_graphs, n, C, cutoff_rate, seed, device):
    """charged N-body systems as the reference simulator initialises them
    (datasets/nbody/datagen/system.py:21,36-39): positions N(0, ((n/5)^(1/3)+0.1)^2), |vel| = 0.5, charges +-1;
    edges = the shortest (1 - cutoff_rate) fraction of the n(n-1) directed pairs (datasets/nbody/dataset.py:102-113);
    node_feat=[|vel|, charge/max], edge_attr=[q_i q_j, dist]; target = loc + 0.4 vel."""
    import torch
    g = torch.Generator().manual_seed(seed)
    scale = (n / 5.0) ** (1.0 / 3.0) + 0.1
    loc = torch.randn(n_graphs, n, 3, generator=g) * scale
    vel = torch.randn(n_graphs, n, 3, generator=g)
    vel = vel * (0.5 / vel.norm(dim=2, keepdim=True))
    q = (torch.randint(0, 2, (n_graphs, n), generator=g) * 2 - 1).float()
    k = int(n * (n - 1) * (1 - cutoff_rate))
    d = torch.cdist(loc, loc) + torch.eye(n) * 1e18
    idc = torch.topk(d.reshape(n_graphs, n * n), k, dim=1, largest=False).indices        # ascending length
    row, col = idc.div(n, rounding_mode="trunc"), idc.remainder(n)
    off = (torch.arange(n_graphs) * n).unsqueeze(1)
    dist = d.reshape(n_graphs, n * n).gather(1, idc)
    qq = q.gather(1, row) * q.gather(1, col)
    N = n_graphs * n
    batch = dict(
        node_feat=torch.stack([vel.norm(dim=2).reshape(N), q.reshape(N)], 1),
        node_loc=loc.reshape(N, 3), node_vel=vel.reshape(N, 3),
        edge_index=torch.stack([(row + off).reshape(-1), (col + off).reshape(-1)]),
        data_batch=torch.arange(n_graphs).repeat_interleave(n),
        loc_mean=loc.mean(1).unsqueeze(-1).repeat(1, 1, C),
        edge_attr=torch.stack([qq.reshape(-1), dist.reshape(-1)], 1),
    )
    target = (loc + 0.4 * vel).reshape(N, 3)
    return {k_: v.contiguous().to(device) for k_, v in batch.items()}, target.to(device)


def make_protein_batch(n_graphs, n, C, cutoff_rate, seed, device, radius=10.0, box=36.0, offset=50.0):
    """graphs of n points uniform in a `box` A cube translated by +offset A, contacts within
    `radius` A (datasets/protein/dataset.py:146) minus the longest cutoff_rate fraction (run_protein.sh:4)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    locs, eis, dists, off = [], [], [], 0
    for b in range(n_graphs):
        loc = torch.rand(n, 3, generator=g) * box + offset
        if str(device) != "cpu":
            from fastegnn_amd.graphs import cutoff_edges, radius_graph
            ei, d = radius_graph(loc.to(device), radius)
            ei, d = cutoff_edges(ei, d, cutoff_rate)
            ei, d = ei.cpu(), d.cpu()
        else:
            dd = torch.cdist(loc, loc) + torch.eye(n) * 1e18
            r, c = torch.nonzero(dd <= radius, as_tuple=True)
            d = dd[r, c]
            order = torch.argsort(d, stable=True)[: int(r.numel() * (1 - cutoff_rate))]
            ei, d = torch.stack([r[order], c[order]]), d[order]
        locs.append(loc); eis.append(ei + off); dists.append(d); off += n
    loc, ei, dist = torch.cat(locs), torch.cat(eis, 1), torch.cat(dists)
    N = n_graphs * n
    vel = torch.randn(N, 3, generator=g) * 0.3
    batch = dict(
        node_feat=torch.stack([vel.norm(dim=1), torch.rand(N, generator=g)], 1),
        node_loc=loc, node_vel=vel, edge_index=ei,
        data_batch=torch.arange(n_graphs).repeat_interleave(n),
        loc_mean=torch.stack([l.mean(0) for l in locs]).unsqueeze(-1).repeat(1, 1, C),
        edge_attr=torch.stack([dist, dist], 1),
    )
    target = loc + 0.5 * vel
    return {k_: v.contiguous().to(device) for k_, v in batch.items()}, target.to(device)


def make_workload(cfg, seed, device, nodes=None, channels=None):
    C = channels or cfg["C"]
    if cfg["kind"] == "water":
        frame, target = make_frame(nodes or cfg["nodes"], C, seed, device)
        if os.environ.get("FASTEGNN_BENCH_NODE_ORDER") == "morton":
            # DIAGNOSTIC ONLY (never the reported configuration): the same frame with its nodes relabelled along a Morton
            # curve -- what spatial locality of the node order is worth to the gather / scatter kernels
            import torch
            from fastegnn_amd.sharded import morton_order
            order = morton_order(frame["node_loc"], frame["data_batch"])
            inv = torch.empty_like(order)
            inv[order] = torch.arange(order.numel(), device=order.device)
            frame = dict(frame, node_feat=frame["node_feat"][order], node_loc=frame["node_loc"][order],
                         node_vel=frame["node_vel"][order], edge_index=inv[frame["edge_index"]])
            target = target[order]
        return frame, target
    if cfg["kind"] == "nbody":
        return make_nbody_batch(cfg["graphs"], nodes or cfg["nodes"], C, cfg["cutoff"], seed, device)
    return make_protein_batch(cfg["graphs"], nodes or cfg["nodes"], C, cfg["cutoff"], seed, device)


def loss_fn(loc, vloc, target):
    import torch
    return torch.nn.functional.mse_loss(loc, target) + 0.01 * vloc.pow(2).mean()


def kernel_model(N, E, B, C, L, gravity=True, phased=False):
    """Per-LAUNCH algorithmic work of each kernel: name -> (FLOPs, bytes or None).

    FLOPs: the 64x64 contractions the stage requires, no tile padding (UNIT = one 64x64 mat-vec).
    bytes: SURVEY.md section 8d only -- the edge-scatter formula 280 E + 540 N (forward; the backward counts twice that,
    'bytes_bwd = 2 bytes_fwd') and the node/virtual figure of 0.82 KB per node and layer (forward; twice for the
    backward).  Kernels for which section 8d states no byte figure get None: their operand arrays exist because of
    implementation choices, and a GB/s figure computed from them is not a roofline number (`operand_bytes` below reports
    what the streaming helper kernels read, separately)."""
    NC = N * C
    heads = 2 if gravity else 1
    f = {}
    f["edge_fwd_kernel"] = (E * 2 * UNIT, 280 * E + 540 * N)
    # edge backward: 2 recomputed + 2 transposed layers + the two in-workgroup weight-gradient contractions per edge
    f["edge_bwd_kernel"] = (E * 6 * UNIT, 2 * (280 * E + 540 * N))
    f["virt_fwd_kernel"] = ((NC * 4 + N * 3) * UNIT, 820 * N)
    # virtual backward (producer/consumer form): 3 recomputed + 3 transposed products + 3 in-workgroup weight gradients
    # per (node, channel); the W3c^T product and the node-MLP adjoint are kernels of their own
    f["virt_bwd_kernel"] = (NC * 9 * UNIT, 2 * 820 * N)
    if phased:
        # the channel-phased form (round 5) contains the W3c^T product and the dW3c contraction as well: 3 recomputed + 4 transposed
        # products + 4 in-workgroup weight gradients per (node, channel); no virt_bwd_gv kernel, no per-channel wgrad job
        f["virt_bwd_kernel"] = (NC * 11 * UNIT, 2 * 820 * N)
    f["virt_bwd_gv_kernel"] = (NC * UNIT, None)
    f["virt_bwd_node_kernel"] = (N * 3 * UNIT, None)
    f["node_pre_fwd_kernel"] = (N * (3 + heads) * UNIT, None)
    f["node_pre_bwd_kernel"] = (N * (3 + 2 * heads) * UNIT, None)
    return f


def recompute_free_units(N, E, B, C, phased=False):
    """FLOPs per launch WITHOUT the forward products a backward kernel recomputes (SURVEY 8d's convention: backward = 2 x
    forward -- the transposed products and the weight-gradient contractions): what `roofline.frac_algorithmic` is computed
    from, so that recomputation shows up as lost efficiency instead of as achieved FLOPs."""
    NC = N * C
    return {"edge_bwd_kernel": E * 4 * UNIT, "virt_bwd_kernel": NC * (8 if phased else 6) * UNIT}


def operand_bytes(N, E, B, C, gravity=True):
    """Bytes the streaming helper kernels must read per launch (distinct operand arrays, each counted once)."""
    heads = 2 if gravity else 1
    return {
        # per-edge d/d(Q|x) rows + the inverted index, one output row per node
        "edge_col_reduce_kernel": E * (272 + 4) + N * (272 + 4),
        # layer-wide batch: g_h_out t3 g_np h aggm g_P g_A + velocity / gravity operands ([N,64]) and g_QX ([N,68])
        "wgrad_tn_kernel[layer batch]": N * 256 * (7 + heads) + N * 272 + 5 * B * C * 512,
        # node_mlp.0 blocks of the C channels: v [C][N][64] once, g_np [N,64] once
        "wgrad_tn_kernel[v job]": N * 256 * (C + 1),
    }


# ------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (op-for-op restatement of the reference) timed on the host cores
# ------------------------------------------------------------------------------------------------------
def host_cores():
    """(logical CPUs visible to this process, physical cores of the machine if /proc/cpuinfo tells)."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    phys = None
    try:
        seen = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    seen.add((pid, cid))
                pid = cid = None
        phys = len(seen) or None
    except OSError:
        pass
    return logical, phys


def cpu_baseline(cfg, args, seed, mode):
    """-> dict for the JSON line.  mode 'full': ONE fwd+bwd of the oracle on the full workload after a small
    warm-up pass; 'scaled': median of 3 on a bounded sample of the same density, scaled linearly (flattering to the
    CPU: the reference is super-linear in N)."""
    import numpy as np
    import torch
    from oracle import fastegnn_ref as R
    logical, phys = host_cores()
    C = args.channels or cfg["C"]
    ocfg = R.Config(node_feat_nf=2, node_attr_nf=0, edge_attr_nf=2, hidden_nf=64, virtual_channels=C,
                    n_layers=args.layers, gravity=cfg["gravity"])
    p = {k: v.requires_grad_(True) for k, v in R.init_params(ocfg, seed=43).items()}

    def run(frame, target):
        for v in p.values():
            v.grad = None
        t0 = time.perf_counter()
        loc, vloc = R.forward(p, ocfg, **frame)
        loss_fn(loc, vloc, target).backward()
        return time.perf_counter() - t0

    # thread count: --cpu-threads, else the fastest of {8,16,32,64} on a 4000-node frame (this op mix -- gathers,
    # scatter_add_, small GEMMs -- does not scale with threads: measured 3.2 s at 8 vs 20 s at 256 threads on a 10 k frame)
    cal = None
    if args.cpu_threads:
        threads = max(1, min(args.cpu_threads, logical))
    else:
        # (more than 64 threads is never faster here and costs minutes: 96 s per pass at 256 threads on the 4000-node frame)
        cand = sorted({t for t in (8, 16, 32, 64) if t <= logical} or {logical})
        fr = make_frame(4000, C, seed, "cpu")
        cal = {}
        for t in cand:
            torch.set_num_threads(t)
            run(*fr)
            cal[t] = round(min(run(*fr), run(*fr)), 3)
        threads = min(cal, key=cal.get)
    torch.set_num_threads(threads)
    full_nodes = args.nodes or cfg["nodes"]
    if mode == "full":
        if cfg["kind"] == "water":
            run(*make_frame(2000, C, seed, "cpu"))                       # page the code in
        frame, target = make_workload(cfg, seed, "cpu", args.nodes, args.channels)
        t = run(frame, target)
        e = frame["edge_index"].size(1)
        units = cfg.get("graphs", 1)
        sample = (f"oracle/fastegnn_ref.py (torch CPU, op-for-op restatement of the reference) fwd+bwd, ONE pass over the "
                  f"full workload ({frame['node_loc'].size(0)} nodes, {e} edges, C={C}) after a 2000-node warm-up pass: {t:.2f} s")
        value = units / t
    else:
        ns = min(full_nodes, 10000)
        frame, target = make_frame(ns, C, seed, "cpu")
        ts = [run(frame, target) for _ in range(4)]
        t = float(np.median(ts[1:]))
        scale = full_nodes / ns
        sample = (f"oracle/fastegnn_ref.py fwd+bwd on a {ns}-node frame of the same density (E={frame['edge_index'].size(1)}), "
                  f"median of 3 = {t:.2f} s, scaled x{scale:.0f} linearly to {full_nodes} nodes (the reference is "
                  f"super-linear in N, so this flatters the CPU)")
        value = 1.0 / (t * scale)
    return {"value": round(value, 5), "unit": "graphs/s", "cores": threads, "kind": "port", "sample": sample,
            "host_logical_cpus": logical, "host_physical_cores": phys, "torch_threads": threads,
            "thread_calibration_s_4000_nodes": cal}


def latest_traffic():
    """HBM bytes per launch from the newest rocprofv3 PMC capture under profiles/ (tools/gpu_traffic.sh)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    if not files:
        return {}, None
    return json.load(open(files[-1])), os.path.basename(files[-1])


def latest_sq_counters():
    """Per-launch SQ counters of the fused kernels from the newest tracked capture (tools/gpu_sq.sh -> profiles/sq_counters_rNN.json):
    what the matrix pipe really executed (SQ_INSTS_VALU_MFMA_MOPS_*: 512 FLOP per count) and how busy it was."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "sq_counters_r*.json")))
    if not files:
        return {}, None
    return json.load(open(files[-1])), os.path.basename(files[-1])


# profiler ids (fastegnn_profile_name) -> kernel names of the counter capture
_SQ_NAMES = {"virt_bwd_kernel": "fe::virt_bwd_pc_kernel", "edge_bwd_kernel": "fe::edge_bwd_pc_kernel", "edge_fwd_kernel": "fe::edge_fwd_kernel",
             "virt_fwd_kernel": "fe::virt_fwd_kernel", "node_pre_fwd_kernel": "fe::node_pre_fwd_kernel",
             "node_pre_bwd_kernel": "fe::node_pre_bwd_kernel", "virt_bwd_node_kernel": "fe::virt_bwd_node_kernel",
             "virt_bwd_gv_kernel": "fe::virt_bwd_gv_kernel"}


def matrix_pipe(sq, name, launch_s):
    """(matrix_pipe_util, mfma_busy) of a kernel: the 16-bit MFMA work the counters saw per launch / this run's launch duration /
    the dense 16-bit peak, and the tracked busy share of the matrix pipe.  None when the capture has no such kernel."""
    c = sq.get(_SQ_NAMES.get(name, name))
    if name == "virt_bwd_kernel" and sq.get("fe::virt_bwd_cs_kernel"):   # the channel-phased form of the same stage (csrc/virt_bwd.hip)
        c = sq["fe::virt_bwd_cs_kernel"]
    if not c:
        return None, None
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    util = mops * 512.0 / launch_s / (PEAK_MFMA_BF16_TFLOPS * 1e12)
    return round(util, 4), (round(c["mfma_busy"], 4) if c.get("mfma_busy") is not None else None)


# ------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg4")
    ap.add_argument("--nodes", type=int, default=None, help="override the node count (per graph for cfg1..cfg3)")
    ap.add_argument("--channels", type=int, default=None, help="override the number of virtual channels")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--mode", choices=["auto", "sharded", "dp"], default="auto",
                    help="N>1: 'sharded' = ONE batch partitioned over the ranks (strong scaling; default for cfg4/cfg5), "
                         "'dp' = one batch per rank + gradient all-reduce (weak scaling; default for cfg1..cfg3)")
    ap.add_argument("--sharded", action="store_true", help="same as --mode sharded (also at 1 GPU: the staged C entry points)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="1 GPU only: this process plays ONE rank (--emulate-rank) of a W-rank partition of the frame -- its rows and "
                         "edges, the interior / boundary split, the pack / unpack kernels and byte counts of its halo, device copies "
                         "in the place of the xGMI transfers: the compute and fixed costs of a 1/W shard, measured (implies --sharded; "
                         "`value` is then the rate of THAT shard's step, an upper bound of the W-GPU job's rate)")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--dtype", choices=["auto", "f32", "bf16"], default="auto",
                    help="MLP operand type: bf16 = bf16 operands / fp32 accumulate (default for cfg3), f32 otherwise")
    ap.add_argument("--cpu-baseline", choices=["auto", "full", "scaled", "none"], default="auto",
                    help="auto: 'full' at 1 GPU for cfg1..cfg4 (one oracle pass over the whole workload), 'scaled' for cfg5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU baseline (default: all visible CPUs)")
    ap.add_argument("--cache-graph", action="store_true", help="reuse the sorted graph across steps")
    ap.add_argument("--hipgraph", choices=["auto", "on", "off"], default="auto",
                    help="replay the whole step (CSR build + forward + loss + backward) as ONE captured HIP graph: removes the "
                         "~100 launch latencies per step that bound the small configurations; auto = on for cfg1/cfg2/cfg3 at 1 GPU")
    ap.add_argument("--train-step", action="store_true",
                    help="time a full training iteration instead (edge_attr augmentation, MSE+MMD loss, Adam: "
                         "fastegnn_amd.train.train_step); the default step is fwd+loss+bwd, the BASELINE metric")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # before any GPU / HIP call in this process

    import numpy as np  # noqa: F401
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # FASTEGNN_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (the ranks share the
    # visible devices and the collectives are staged through host memory; RCCL refuses two ranks on one device)
    backend = os.environ.get("FASTEGNN_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    import fastegnn_amd
    from fastegnn_amd import _lib as K
    from fastegnn_amd.dist import allreduce_gradients, init_from_env, max_over_ranks
    init_from_env(backend)

    cfg = CONFIGS[args.config]
    C, L = args.channels or cfg["C"], args.layers
    mode = args.mode
    if args.emulate_world:
        if world != 1:
            raise SystemExit("--emulate-world needs --gpus 1")
        args.sharded = True
    if args.sharded:
        mode = "sharded"
    if mode == "auto":
        mode = "sharded" if (world > 1 and cfg["kind"] == "water") else "dp"
    sharded = mode == "sharded"
    dtype = cfg["dtype"] if args.dtype == "auto" else args.dtype

    torch.manual_seed(43)
    extra = {}
    if dtype == "bf16":
        extra["mlp_dtype"] = torch.bfloat16
    model = fastegnn_amd.FastEGNN(node_feat_nf=2, node_attr_nf=0, edge_attr_nf=2, hidden_nf=64, virtual_channels=C,
                                  device=dev, n_layers=L, gravity=cfg["gravity"], **extra)
    model.cache_graphs = bool(args.cache_graph)
    # sharded: every rank builds the same batch and keeps its own rows / edges; dp: a different batch per rank
    frame, target = make_workload(cfg, 43 + (0 if sharded else rank), dev, args.nodes, args.channels)
    N, E, B = frame["node_loc"].size(0), frame["edge_index"].size(1), frame["loc_mean"].size(0)
    params = [p for p in model.parameters()]

    if args.train_step:
        from fastegnn_amd.train import FusedAdam, train_step
        opt = FusedAdam(params, lr=5e-4, weight_decay=1e-12)
        gsel = torch.Generator().manual_seed(7)
        samp = torch.randperm(N, generator=gsel)[: 3 * C].view(1, -1).to(dev)
        data = dict(loc_0=frame["node_loc"], vel_0=frame["node_vel"], loc_t=target, node_feat=frame["node_feat"],
                    edge_index=frame["edge_index"], edge_attr=frame["edge_attr"][:, :1].contiguous(),
                    batch=frame["data_batch"], loc_mean=frame["loc_mean"])

    stats = None
    if sharded:
        from fastegnn_amd.sharded import CommStats, ShardedFastEGNN
        smodel = ShardedFastEGNN(model, emulate=(args.emulate_world, args.emulate_rank) if args.emulate_world else None)
        # nodes sorted along a Morton curve first (a Water-3D loader would do this once per frame): contiguous index
        # ranges are compact regions, so the halo exchange moves the shell of a region, not the whole table
        shard = smodel.shard_inputs(**frame, reorder=True)   # this rank's rows and edges; the full COO is dropped below
        tgt_local = shard["plan"].rows(target)
        if world > 1:
            del frame
            torch.cuda.empty_cache()
            # second leg of the N > 1 line: graphs as independent units (one frame per rank + gradient all-reduce)
            dp_frame, dp_target = make_workload(cfg, 143 + rank, dev, args.nodes, args.channels)

    def step():
        if args.train_step:   # utils/train.py:30-170 on device (single-GPU only)
            return train_step(model, opt, data, samp, 1.0, 0.01)[0]
        for p in params:
            p.grad = None
        if sharded:
            # this rank's rows of the MSE; the virtual-node term is replicated IN FULL on every rank (the virtual
            # state is replicated: fastegnn_amd/sharded.py 'Loss contract'), so the printed loss of rank r is
            # mse_r + the full vloc term
            loc, vloc = smodel.forward_local(shard)
            loss = (loc - tgt_local).pow(2).sum() / (3 * N) + 0.01 * vloc.pow(2).mean()
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

    # auto: every single-GPU configuration up to cfg4's size replays its step as one captured HIP graph (cfg4: 15.5 -> 15.0 ms,
    # the host-side launch gaps of ~110 launches); cfg5 gains nothing (212 ms either way) and its capture pool would pin
    # ~50 GB, so it stays eager
    use_graph = args.hipgraph == "on" or (args.hipgraph == "auto" and world == 1 and not sharded and not args.train_step
                                          and (args.nodes or cfg["nodes"]) <= 200000)
    # the per-kernel HIP-event profiler is on during warm-up too, so that its event pool exists
    # before the timed region (hipEventCreate is slow on a cold driver)
    K.lib().fastegnn_profile_enable(1)
    # two extra untimed steps ahead of the W warm-up steps: on a freshly booted box the first pass still pages
    # the Python / torch / HIP code in from the image, which showed up as sporadic 100 ms host stalls
    for _ in range(2 + max(args.warmup, 1)):
        step()
    sync()
    K.profile_collect()
    prof_steps = args.steps
    prof_taken = False
    eager_ms = None
    if use_graph:
        # per-kernel durations from an eager pass (events cannot be timed inside a captured graph), then the step is
        # captured once and the timed region replays it
        prof_steps = min(args.steps, 20)
        t0 = time.perf_counter()
        for _ in range(prof_steps):
            step()
        sync()
        eager_ms = (time.perf_counter() - t0) / prof_steps * 1e3      # eager launches, per-kernel events on (slightly pessimistic)
        K.lib().fastegnn_profile_enable(0)
        prof = K.profile_collect()
        prof_taken = True
        # the same eager step without the event pairs around every kernel: what a training loop over frames with varying
        # edge counts pays (a captured graph cannot be replayed for a different E)
        t0 = time.perf_counter()
        for _ in range(prof_steps):
            step()
        sync()
        eager_ms = min(eager_ms, (time.perf_counter() - t0) / prof_steps * 1e3)
        gstream = torch.cuda.Stream(dev)
        gstream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(gstream):
            step()                                   # allocator warm-up on the capture stream
        torch.cuda.current_stream(dev).wait_stream(gstream)
        hgraph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(hgraph, stream=gstream):
                loss = step()
        except Exception as exc:                      # a box whose runtime cannot capture the step: time eager launches
            print(f"bench: HIP graph capture failed ({type(exc).__name__}: {exc}); timing eager launches", file=sys.stderr)
            torch.cuda.synchronize(dev)
            use_graph, hgraph = False, None
    if use_graph:
        for _ in range(max(args.warmup, 1)):
            hgraph.replay()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hgraph.replay()
        sync()
        dt = time.perf_counter() - t0
    else:
        if sharded:
            stats = CommStats()
            smodel.stats = stats
        t0 = time.perf_counter()
        done = []   # bound the CPU run-ahead to two steps (deep HIP queues stall sporadically on this stack)
        for _ in range(args.steps):
            if len(done) >= 2:
                done.pop(0).synchronize()
            loss = step()
            ev = torch.cuda.Event()
            ev.record()
            done.append(ev)
        sync()
        dt = time.perf_counter() - t0
        if not prof_taken:   # (after a failed graph capture the per-kernel table of the eager pre-pass is kept)
            K.lib().fastegnn_profile_enable(0)
            prof = K.profile_collect()
    dt = max_over_ranks(dt, dev)

    dp_leg = None
    if sharded and world > 1:
        # the same job as data-parallel replicas (weak scaling): every rank its own frame, no data-path collective,
        # one flat gradient all-reduce -- reported beside the partitioned number, same steps / warm-up / barriers
        def dp_step():
            for p in params:
                p.grad = None
            loc, vloc = model(**dp_frame)
            l = loss_fn(loc, vloc, dp_target)
            l.backward()
            allreduce_gradients(params)
            return l
        for _ in range(max(args.warmup, 1)):
            dp_step()
        sync()
        t0 = time.perf_counter()
        done = []
        for _ in range(args.steps):
            if len(done) >= 2:
                done.pop(0).synchronize()
            dp_step()
            ev = torch.cuda.Event()
            ev.record()
            done.append(ev)
        sync()
        dt_dp = max_over_ranks(time.perf_counter() - t0, dev)
        dp_leg = {"value": round(world * B * args.steps / dt_dp, 4), "unit": "graphs/s", "ms_per_step": round(dt_dp / args.steps * 1e3, 3),
                  "scaling": "weak", "parallelism": f"dp{world}: one frame per GPU, RCCL gradient all-reduce (2.2 MB)"}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        units_per_step = B * (1 if sharded else world)          # graphs the whole job completes per step
        value = units_per_step * args.steps / dt
        # kernel model of what THIS rank ran (sharded: its 1/world share of the rows and edges)
        kN, kE = (shard["plan"].nloc, shard["edge_index"].size(1)) if sharded else (N, E)
        phased = "virt_bwd_kernel" in prof and "virt_bwd_gv_kernel" not in prof      # which form of the virtual backward ran
        km = kernel_model(kN, kE, B, C, L, gravity=cfg["gravity"] is not None, phased=phased)
        ob = operand_bytes(kN, kE, B, C, gravity=cfg["gravity"] is not None)
        kernels = {}
        rfree = recompute_free_units(kN, kE, B, C, phased=phased)
        sq, sq_file = latest_sq_counters() if (args.config == "cfg4" and not sharded and dtype != "bf16") else ({}, None)
        sq_meta = sq.get("_meta", {}) if sq else {}
        for name, (ms, cnt) in prof.items():
            per_step = ms / prof_steps
            launch_s = ms / cnt * 1e-3
            ent = {"ms_per_step": round(per_step, 4), "launches_per_step": cnt / prof_steps,
                   "avg_launch_ms": round(ms / cnt, 5)}
            if name in km:
                fl, by = km[name]
                # fp32-EQUIVALENT rate: algorithmic 64x64 products per launch (recomputed ones included) x 8192 FLOP / launch duration.
                # The products run as 16-bit MFMAs (f16x2 / bf16x3 splits): this is not a fraction of any pipe's peak and may exceed
                # the 157.3 TFLOP/s of the fp32-input MFMA; what the matrix pipe did is `matrix_pipe_util`
                ent["fp32_equivalent_tflops"] = round(fl / launch_s / 1e12, 3)
                if by is not None:
                    ent["gbs"] = round(by / launch_s / 1e9, 1)                   # SURVEY 8d algorithmic bytes per launch
                util, busy = matrix_pipe(sq, name, launch_s)
                if util is not None:
                    ent["matrix_pipe_util"] = util
                    ent["mfma_busy"] = busy
            if name == "edge_col_reduce_kernel":
                ent["operand_gbs"] = round(ob[name] / launch_s / 1e9, 1)
            if name == "wgrad_tn_kernel":
                if phased:   # ONE launch per layer: the layer-wide batch (dW3c goes through the phased kernel's own slabs + wgrad_reduce)
                    ent["operand_gbs"] = round(ob["wgrad_tn_kernel[layer batch]"] / launch_s / 1e9, 1)
                else:        # tile-major form: two launches per layer, the second is the per-channel node_mlp.0 job over v
                    tot = ob["wgrad_tn_kernel[layer batch]"] + ob["wgrad_tn_kernel[v job]"]
                    ent["operand_gbs"] = round(tot / (2 * launch_s) / 1e9, 1)
            kernels[name] = ent
        dom = max((n for n in kernels if n in km), key=lambda n: kernels[n]["ms_per_step"])
        fl, by = km[dom]
        fl_alg = rfree.get(dom, fl)            # without the forward products a backward kernel recomputes (bwd = 2 x fwd)
        launch_s = kernels[dom]["avg_launch_ms"] * 1e-3
        peak_mfma = PEAK_MFMA_BF16_TFLOPS if dtype == "bf16" else PEAK_MFMA_F32_TFLOPS
        t_mfma = fl / (peak_mfma * 1e12)
        t_hbm = (by or 0) / (PEAK_HBM_GBS * 1e9)
        tr, tr_file = latest_traffic()
        traffic = tr.get(dom, {}).get("hbm_bytes_per_launch") if (args.config == "cfg4" and not sharded) else None
        if t_mfma >= t_hbm:
            # achieved = ALGORITHMIC FLOPs (the transposed products and the weight-gradient contractions of a backward
            # kernel, NOT its recomputed forward products) per launch / the launch duration measured in this run; peak = the dense
            # MFMA peak of the arithmetic type (fp32: 157.3 TFLOP/s).  Recomputation and the 16-bit split products therefore show up
            # as lost efficiency, never as achieved work.
            ach = fl_alg / launch_s / 1e12
            # `frac` is a FP32-EQUIVALENT figure (algorithmic fp32 FLOPs over the fp32-input MFMA peak), NOT the utilisation of the pipe the
            # kernel runs on: that is `matrix_pipe_util` right beside it
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 3), "peak": peak_mfma,
                    "unit": "TFLOP/s", "frac": round(ach / peak_mfma, 4), "frac_is": "fp32-equivalent algorithmic FLOPs / fp32-input MFMA peak; "
                    "the 16-bit matrix pipe's own utilisation is matrix_pipe_util", "traffic": traffic,
                    "algorithmic_flops_per_launch": fl_alg, "executed_product_flops_per_launch": fl}
            util, busy = matrix_pipe(sq, dom, launch_s)
            roof["matrix_pipe_util"] = util      # 16-bit MFMA FLOPs the counters saw / duration / 2.5 PFLOP/s dense
            roof["mfma_busy"] = busy             # SQ_VALU_MFMA_BUSY_CYCLES / the SIMDs' time, from the tracked capture
            roof["counters_source"] = (f"profiles/{sq_file}: rocprofv3 --pmc passes (tools/gpu_sq.sh) of commit "
                                       f"{sq_meta.get('commit', '?')}, not of this run; durations are this run's") if util is not None else None
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["gbs"], "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(kernels[dom]["gbs"] / PEAK_HBM_GBS, 4), "traffic": traffic}
        roof["traffic_source"] = (f"profiles/{tr_file}: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE) of the commit named in "
                                  f"that file, not of this run") if traffic is not None else None
        es = kernels.get("edge_fwd_kernel")
        edge_scatter = None
        if es:
            edge_scatter = {"kernel": "edge_fwd_kernel", "algorithmic_bytes_per_launch": 280 * kE + 540 * kN,
                            "achieved_GBs": es["gbs"], "frac_of_hbm_peak": round(es["gbs"] / PEAK_HBM_GBS, 4),
                            "fp32_equivalent_tflops": es["fp32_equivalent_tflops"],
                            "matrix_pipe_util": es.get("matrix_pipe_util"), "mfma_busy": es.get("mfma_busy")}
        if sharded:
            par = f"row-owner sharding of ONE batch over {world} GPU(s) (fastegnn_amd/sharded.py), strong scaling"
        elif world > 1:
            par = f"dp{world}: one batch per GPU, RCCL gradient all-reduce, weak scaling"
        else:
            par = "1 GPU"
        what = "full training iteration (augment+fwd+MSE/MMD+bwd+Adam)" if args.train_step else "fwd+loss+bwd"
        out = {
            "metric": "graphs/sec (fwd+bwd), " + ("Water-3D-like 100k-node frame" if args.config == "cfg4" else args.config),
            "value": round(value, 4),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "eager_ms_per_step": round(eager_ms if eager_ms is not None else ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{cfg['text']}; {L}-layer FastEGNN H=64, {what}, CSR build "
                                   + ("cached" if args.cache_graph else "inside the step"),
                       "name": args.config, "nodes": N, "edges": E, "graphs_per_batch": B, "virtual_channels": C, "layers": L,
                       "graphs_per_step": units_per_step, "parallelism": par, "loss": float(loss.detach()),
                       "launch": ("whole step replayed as one captured HIP graph; per-kernel durations from an eager pass of "
                                  f"{prof_steps} steps before the capture") if use_graph else "eager launches"},
            "value_per_gpu": round(value / world, 4),
            "peak_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
            "roofline": roof,
            "edge_scatter": edge_scatter,
            "kernels": kernels,
        }
        if stats is not None:
            out["collectives"] = stats.summary(args.steps)
        if sharded:
            pl = shard["plan"]
            out["launches_per_step"] = sum(v["launches_per_step"] for v in kernels.values())
            out["shard"] = {"rows": pl.nloc, "edges": kE, "edge_stage_launch_ranges": [
                {"first_row": r0, "rows": n, "waits_for_halo": bool(h)} for r0, n, h in pl.parts],
                "transport": os.environ.get("FASTEGNN_COMM", "torch"),
                "emulated_rank_of_world": [args.emulate_rank, args.emulate_world] if args.emulate_world else None}
            if args.emulate_world:
                out["metric"] += f" -- ONE emulated rank of {args.emulate_world} (its shard's step on one GPU, no xGMI transfers)"
            out["table_exchange"] = {"mode": pl.mode, "rows_received_per_exchange": pl.exchanged_bytes() // (68 * 4),
                                     "bytes_received_per_exchange": pl.exchanged_bytes(),
                                     "all_gather_bytes_for_comparison": (world - 1) * pl.Npad * 68 * 4,
                                     "node_order": "Morton curve inside each graph (fastegnn_amd.sharded.morton_order)",
                                     "collective_schedule": ("blocking, program order (default until RCCL with N > 1 has been "
                                                             "verified; FASTEGNN_SHARDED_SYNC=0 overlaps)"
                                                             if os.environ.get("FASTEGNN_SHARDED_SYNC", "1") not in ("", "0")
                                                             else "asynchronous, waited for at the first consumer")}
        if dp_leg is not None:
            out["data_parallel_leg"] = dp_leg
        cb = "none" if args.no_cpu_baseline else args.cpu_baseline
        if cb == "auto":
            cb = "none" if world > 1 else ("scaled" if args.config == "cfg5" else "full")
        if cb != "none" and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args, 43, cb)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)   # flushed before any communicator teardown
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
