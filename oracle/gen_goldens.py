#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden-vector generator.

Runs ONLY in the build container, where the real reference is mounted at
/root/reference.  It imports ``models/FastEGNN.py`` from there (with a stand-in
for the one third-party symbol it needs, ``torch_geometric.nn.global_mean_pool``,
torch_geometric==2.5.2 being absent), runs forward + autograd backward on small
seeded cases and stores plain arrays under ``tests/golden/*.npz``:

    in/<name>      inputs
    p/<name>       state_dict tensors (reference key names)
    out/loc, out/vloc, out/layer<i>/{h,x,Hv,Z}
    gp/<name>      d loss / d parameter
    gin/<name>     d loss / d {node_feat,node_loc,node_vel,loc_mean}
    meta/*         constructor flags, loss definition

No reference source, bytecode or pickled class leaves /root/reference; the
fixtures are data only.  Usage:  python oracle/gen_goldens.py
"""
import os
import sys
import types
import random

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_pyg_stand_in():
    """Per-graph mean with PyG 2.5.2 semantics (size = batch.max()+1, empty -> 0),
    written as an explicit loop so that it is independent of the oracle's helper."""
    def global_mean_pool(x, batch, size=None):
        n = int(batch.max()) + 1 if size is None else size
        rows = []
        for b in range(n):
            sel = x[batch == b]
            rows.append(sel.mean(0) if sel.size(0) > 0 else x.new_zeros(x.size(1)))
        return torch.stack(rows, 0)

    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgnn.global_mean_pool = global_mean_pool
    tg.nn = tgnn
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = tgnn


def _import_reference():
    _install_pyg_stand_in()
    sys.path.insert(0, REF)
    from models.FastEGNN import FastEGNN  # noqa
    return FastEGNN


def _rand_graph_batch(gen, sizes, edges_per_graph, self_loops=True, isolate=None):
    """Ragged batch: random directed edges inside each graph (duplicates allowed)."""
    rows, cols, batch = [], [], []
    off = 0
    for g, (n, e) in enumerate(zip(sizes, edges_per_graph)):
        r = torch.randint(0, n, (e,), generator=gen)
        c = torch.randint(0, n, (e,), generator=gen)
        if not self_loops:
            c = torch.where(c == r, (c + 1) % n, c)
        if isolate is not None and isolate[0] == g:
            k = isolate[1]
            r = torch.where(r == k, (r + 1) % n, r)   # node k never aggregates
        rows.append(r + off)
        cols.append(c + off)
        batch += [g] * n
        off += n
    ei = torch.stack([torch.cat(rows), torch.cat(cols)]).long()
    perm = torch.randperm(ei.size(1), generator=gen)  # datasets emit edges unsorted by row
    return ei[:, perm], torch.tensor(batch, dtype=torch.long)


def _loc_mean(loc, batch, C):
    B = int(batch.max()) + 1
    cm = torch.stack([loc[batch == b].mean(0) for b in range(B)])   # [B,3]
    return cm.unsqueeze(-1).repeat(1, 1, C)


def run_case(FastEGNN, name, *, sizes, edges, nf, na, ea, C, H=64, L=4, seed=0,
             attention=False, normalize=False, tanh=False, gravity=None, residual=True,
             coord_scale=1.0, loc_scale=3.0, isolate=None, dtype=torch.float32,
             explicit=None, act=None, act_param=0.0):
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    gen = torch.Generator().manual_seed(seed + 1000)
    act_kw = {}
    if act is not None:   # the reference constructor's act_fn (models/FastEGNN.py:227); default nn.SiLU()
        act_kw["act_fn"] = {"relu": lambda: torch.nn.ReLU(), "leaky_relu": lambda: torch.nn.LeakyReLU(act_param),
                            "tanh": lambda: torch.nn.Tanh(), "sigmoid": lambda: torch.nn.Sigmoid(),
                            "elu": lambda: torch.nn.ELU(act_param), "gelu": lambda: torch.nn.GELU(),
                            "softplus": lambda: torch.nn.Softplus(beta=act_param)}[act]()
    model = FastEGNN(node_feat_nf=nf, node_attr_nf=na, edge_attr_nf=ea, hidden_nf=H,
                     virtual_channels=C, device="cpu", n_layers=L, residual=residual,
                     attention=attention, normalize=normalize, tanh=tanh, gravity=gravity, **act_kw)
    with torch.no_grad():
        for k, v in model.named_parameters():
            if k.endswith("coord_mlp_r.2.weight") or k.endswith("coord_mlp_r_virtual.2.weight") \
                    or k.endswith("coord_mlp_v_virtual.2.weight"):
                v.mul_(coord_scale)
    model = model.to(dtype)

    if explicit is not None:
        inp = {k: v.to(dtype) if v.is_floating_point() else v for k, v in explicit.items()}
        ei, batch = inp["edge_index"], inp["data_batch"]
        N = inp["node_loc"].size(0)
    else:
        ei, batch = _rand_graph_batch(gen, sizes, edges, isolate=isolate)
        N = batch.numel()
        loc = (torch.randn(N, 3, generator=gen) * loc_scale).to(dtype)
        inp = dict(
            node_feat=torch.rand(N, nf, generator=gen).to(dtype),
            node_loc=loc,
            node_vel=(torch.randn(N, 3, generator=gen) * 0.5).to(dtype),
            edge_index=ei, data_batch=batch,
            loc_mean=_loc_mean(loc, batch, C),
            edge_attr=torch.rand(ei.size(1), ea, generator=gen).to(dtype),
        )
        if na > 0:
            inp["node_attr"] = torch.rand(N, na, generator=gen).to(dtype)
    B = int(batch.max()) + 1
    target = (inp["node_loc"] + torch.randn(N, 3, generator=gen).to(dtype))
    wv = torch.randn(B, 3, C, generator=gen).to(dtype)

    leaf = {}
    for k in ("node_feat", "node_loc", "node_vel", "loc_mean"):
        leaf[k] = inp[k].clone().requires_grad_(True)

    # per-layer intermediates via forward hooks on the gcl_i modules
    layer_out = {}
    hooks = []
    for i in range(L):
        def mk(i):
            def hook(_m, _a, out):
                layer_out[i] = [o.detach().clone() for o in out]
            return hook
        hooks.append(model._modules[f"gcl_{i}"].register_forward_hook(mk(i)))

    loc_pred, vloc = model(node_feat=leaf["node_feat"], node_loc=leaf["node_loc"],
                           node_vel=leaf["node_vel"], edge_index=ei, data_batch=batch,
                           loc_mean=leaf["loc_mean"], edge_attr=inp["edge_attr"],
                           node_attr=inp.get("node_attr"))
    for h in hooks:
        h.remove()
    # loss touching both outputs (the harness's MSE + a linear probe of the virtual coords)
    loss = torch.nn.functional.mse_loss(loc_pred, target) + 0.3 * (vloc * wv).sum() / vloc.numel()
    loss.backward()

    rec = {}
    for k, v in inp.items():
        rec[f"in/{k}"] = v.numpy()
    rec["in/target"] = target.numpy()
    rec["in/wv"] = wv.numpy()
    for k, v in model.state_dict().items():
        rec[f"p/{k}"] = v.detach().numpy()
    rec["out/loc"] = loc_pred.detach().numpy()
    rec["out/vloc"] = vloc.detach().numpy()
    rec["out/loss"] = np.array(loss.item())
    for i in range(L):
        # E_GCL_vel.forward returns (node_feat, coord, virtual_node_feat, virtual_coord)
        h_, x_, Hv_, Z_ = layer_out[i]
        rec[f"out/layer{i}/h"] = h_.numpy()
        rec[f"out/layer{i}/x"] = x_.numpy()
        rec[f"out/layer{i}/Hv"] = Hv_.numpy()
        rec[f"out/layer{i}/Z"] = Z_.numpy()
    for k, v in model.named_parameters():
        rec[f"gp/{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
    for k, v in leaf.items():
        rec[f"gin/{k}"] = v.grad.numpy()
    meta = dict(nf=nf, na=na, ea=ea, C=C, H=H, L=L, attention=int(attention),
                normalize=int(normalize), tanh=int(tanh), residual=int(residual),
                has_gravity=int(gravity is not None))
    for k, v in meta.items():
        rec[f"meta/{k}"] = np.array(v)
    rec["meta/gravity"] = np.array(gravity if gravity is not None else [0, 0, 0], dtype=np.float64)
    if act is not None:
        rec["meta/act"] = np.array(act)
        rec["meta/act_param"] = np.array(float(act_param))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: N={N} E={ei.size(1)} B={B} loss={loss.item():.6f} -> {os.path.getsize(path)/1024:.0f} KiB")


def nbody_case(n_systems=6, n_balls=5, C=3, cutoff_rate=0.5, seed=43):
    """cfg-1 inputs from the reference's own simulator (datasets/nbody/datagen) and the
    dataset's edge/feature construction (datasets/nbody/dataset.py:87-113), restated."""
    sys.path.insert(0, os.path.join(REF, "datasets", "nbody", "datagen"))
    from system import System
    np.random.seed(seed)
    locs, vels, chs = [], [], []
    for _ in range(n_systems):
        s = System(n_isolated=n_balls, n_stick=0, n_hinge=0)
        X, V = [], []
        for t in range(4100):
            s.simulate_one_step()
            if t % 100 == 0:
                X.append(s.X.copy()); V.append(s.V.copy())
        locs.append(np.array(X)); vels.append(np.array(V)); chs.append(s.charges.copy())
    node_feat, loc0, vel0, loct, eis, eas, lms, batch = [], [], [], [], [], [], [], []
    off = 0
    for g in range(n_systems):
        l0 = torch.tensor(locs[g][30], dtype=torch.float32)
        lt = torch.tensor(locs[g][40], dtype=torch.float32)
        v0 = torch.tensor(vels[g][30], dtype=torch.float32)
        ch = torch.tensor(chs[g], dtype=torch.float32)
        n = l0.size(0)
        dist = torch.cdist(l0, l0) + torch.eye(n) * 1e18
        k = int(n * (n - 1) * (1 - cutoff_rate))
        _, idc = torch.topk(dist.view(-1), k, largest=False)
        ei = torch.stack([idc // n, idc % n]).long()
        ea = (l0[ei[0]] - l0[ei[1]]).norm(dim=1, keepdim=True)
        node_feat.append(torch.cat([v0.norm(dim=1, keepdim=True), ch / ch.max()], 1))
        loc0.append(l0); vel0.append(v0); loct.append(lt)
        eis.append(ei + off); eas.append(ea)
        lms.append(l0.mean(0).unsqueeze(-1).repeat(1, C).unsqueeze(0))
        batch += [g] * n
        off += n
    loc0 = torch.cat(loc0); ei = torch.cat(eis, 1)
    ea = torch.cat(eas)
    # harness augmentation (utils/train.py:41-43): [dataset dist || recomputed dist]
    ea2 = torch.cat([ea, (loc0[ei[0]] - loc0[ei[1]]).pow(2).sum(1).sqrt().unsqueeze(1)], 1)
    return dict(node_feat=torch.cat(node_feat), node_loc=loc0, node_vel=torch.cat(vel0),
                edge_index=ei, data_batch=torch.tensor(batch), loc_mean=torch.cat(lms),
                edge_attr=ea2)


def train_case(FastEGNN, name, nb, *, C=3, L=2, seed=43, sigma=1.5, weight=0.01, sample=3, lr=5e-4, wd=1e-12,
               ragged_sizes=None):
    """Row H of SURVEY 8a: the harness's loss (utils/train.py:104-165, MSE + MMD through the reference's
    own `kernel`, :17-20) with the sampled indices fixed, and Adam (main_nbody.py:137) for 3 steps.
    `ragged_sizes` selects the per-graph ('Simulation') branch (:118-142)."""
    stub = types.ModuleType("datasets.protein.dataset")   # utils/train.py:7 imports MDAnalysis through this
    stub.MDAnalysisDataset = object
    sys.modules["datasets.protein.dataset"] = stub
    from utils.train import kernel
    torch.manual_seed(seed)
    model = FastEGNN(node_feat_nf=2, node_attr_nf=0, edge_attr_nf=2, hidden_nf=64, virtual_channels=C, device="cpu",
                     n_layers=L)
    with torch.no_grad():
        for k, v in model.named_parameters():
            if k.endswith(("coord_mlp_r.2.weight", "coord_mlp_r_virtual.2.weight", "coord_mlp_v_virtual.2.weight")):
                v.mul_(100.0)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    gen = torch.Generator().manual_seed(seed + 5)
    N = nb["node_loc"].size(0)
    B = nb["loc_mean"].size(0)
    batch = nb["data_batch"]
    loc_t = nb["node_loc"] + 0.3 * torch.randn(N, 3, generator=gen)
    if ragged_sizes is None:
        n = N // B
        S = min(sample * C, n)
        idx = torch.randperm(n, generator=gen)[:S]
        sample_nodes = (torch.arange(B).unsqueeze(1) * n + idx.unsqueeze(0))          # [B,S] global node ids
    else:
        S = min(sample * C, N)
        rows, off = [], 0
        for nb_ in ragged_sizes:
            rows.append(off + torch.randperm(nb_, generator=gen)[:S])
            off += nb_
        assert all(r.numel() == S for r in rows)
        sample_nodes = torch.stack(rows)

    def loss_of(loc_pred, vloc):
        mse = torch.nn.functional.mse_loss(loc_pred, loc_t)
        V = vloc.permute(0, 2, 1)                                                      # [B,C,3]
        if ragged_sizes is None:   # utils/train.py:144-160
            R_ = loc_pred.reshape(B, -1, 3)[:, idx, :]
            l_vv = torch.sum(kernel(V, V, sigma)) / B / C / C
            l_rv = 2 * torch.sum(kernel(R_, V, sigma)) / B / S / C
        else:                      # utils/train.py:118-142
            l_vv, l_rv = 0.0, 0.0
            for i in range(B):
                Ri = loc_pred[sample_nodes[i]]
                l_vv = l_vv + torch.sum(kernel(V[i], V[i], sigma))
                l_rv = l_rv + torch.sum(kernel(Ri, V[i], sigma))
            l_vv = l_vv / B / C / C
            l_rv = 2 * l_rv / B / S / C
        return mse + weight * (l_vv - l_rv), mse

    rec = {}
    for k, v in nb.items():
        rec[f"in/{k}"] = v.numpy()
    rec["in/loc_t"] = loc_t.numpy()
    rec["in/sample_nodes"] = sample_nodes.numpy().astype(np.int64)
    for k, v in model.state_dict().items():
        rec[f"p0/{k}"] = v.detach().numpy().copy()
    losses = []
    for step in range(1, 4):
        opt.zero_grad()
        loc_pred, vloc = model(node_loc=nb["node_loc"], node_vel=nb["node_vel"], node_attr=None,
                               node_feat=nb["node_feat"], edge_index=nb["edge_index"], loc_mean=nb["loc_mean"],
                               data_batch=batch, edge_attr=nb["edge_attr"])
        loss, mse = loss_of(loc_pred, vloc)
        loss.backward()
        if step == 1:
            rec["out/loc"] = loc_pred.detach().numpy().copy()
            rec["out/vloc"] = vloc.detach().numpy().copy()
            for k, v in model.named_parameters():
                rec[f"g1/{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().copy()
        losses.append([loss.item(), mse.item()])
        opt.step()
        if step in (1, 3):
            for k, v in model.state_dict().items():
                rec[f"p{step}/{k}"] = v.detach().numpy().copy()
    rec["out/losses"] = np.array(losses)
    for k, v in dict(C=C, L=L, sigma=sigma, weight=weight, lr=lr, wd=wd, S=S, ragged=int(ragged_sizes is not None)).items():
        rec[f"meta/{k}"] = np.array(v)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: N={N} B={B} S={S} losses={losses} -> {os.path.getsize(path)/1024:.0f} KiB")


def egnn_cases():
    """Row A13 of SURVEY 8a: the EGNN baseline (models/basic.py:285-341).  basic.py imports torch_sparse /
    torch_scatter / torch_geometric.nn.MessagePassing at module level (:6-8); EGNN never calls them, so inert
    stand-ins are enough."""
    _install_pyg_stand_in()
    sys.modules["torch_geometric.nn"].MessagePassing = type("MessagePassing", (torch.nn.Module,), {})
    for name, attr in (("torch_sparse", "spmm"), ("torch_scatter", "scatter_add")):
        mod = types.ModuleType(name)
        setattr(mod, attr, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stand-in")))
        sys.modules[name] = mod
    sys.path.insert(0, REF)
    from models.basic import EGNN

    def case(name, seed, with_v, coord_scale=1.0, L=2, loc_scale=2.0, norm=False, hidden=64, flat=False):
        torch.manual_seed(seed)
        gen = torch.Generator().manual_seed(seed + 100)
        model = EGNN(n_layers=L, in_node_nf=2, in_edge_nf=2, hidden_nf=hidden, device="cpu", with_v=with_v, norm=norm, flat=flat)
        with torch.no_grad():
            for k, v in model.named_parameters():
                if "coord_net.mlp.2" in k:
                    v.mul_(coord_scale)
        sizes = [7, 4, 9]
        ei, batch = _rand_graph_batch(gen, sizes, [25, 10, 25], isolate=(2, 3))
        N = sum(sizes)
        inp = dict(x=torch.randn(N, 3, generator=gen) * loc_scale, h=torch.rand(N, 2, generator=gen), edge_index=ei,
                   edge_fea=torch.rand(ei.size(1), 2, generator=gen))
        if with_v:
            inp["v"] = torch.randn(N, 3, generator=gen) * 0.5
        leaf = {k: inp[k].clone().requires_grad_(True) for k in ("x", "h") + (("v",) if with_v else ())}
        out = model(**{**inp, **leaf})
        x_out, h_out = out[0], out[-1]
        target = inp["x"] + torch.randn(N, 3, generator=gen)
        wh = torch.randn(N, hidden, generator=gen)
        loss = torch.nn.functional.mse_loss(x_out, target) + 0.05 * (h_out * wh).sum() / N
        loss.backward()
        rec = {f"in/{k}": v.numpy() for k, v in inp.items()}
        rec["in/target"], rec["in/wh"] = target.numpy(), wh.numpy()
        for k, v in model.state_dict().items():
            rec[f"p/{k}"] = v.numpy()
        rec["out/x"], rec["out/h"], rec["out/loss"] = x_out.detach().numpy(), h_out.detach().numpy(), np.array(loss.item())
        for k, v in model.named_parameters():
            rec[f"gp/{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
        for k, v in leaf.items():
            rec[f"gin/{k}"] = v.grad.numpy()
        rec["meta/with_v"], rec["meta/L"], rec["meta/norm"] = np.array(int(with_v)), np.array(L), np.array(int(norm))
        rec["meta/flat"], rec["meta/hidden"] = np.array(int(flat)), np.array(hidden)
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **rec)
        clamped = float((x_out.detach() - inp["x"]).abs().max())
        print(f"{name}: N={N} E={ei.size(1)} loss={loss.item():.5f} max|dx|={clamped:.2f} -> {os.path.getsize(path)/1024:.0f} KiB")

    if "--egnn-wide" in sys.argv:   # the unfused wide path of the sibling: flat=True (Tanh MLPs of 4 x hidden, basic.py:176-178), hidden > 64
        case("egnn_flat", 25, True, coord_scale=3.0, hidden=32, flat=True)
        case("egnn_h128", 26, True, coord_scale=3.0, hidden=128)
        return
    case("egnn_with_v", 21, True)
    case("egnn_no_v", 22, False)
    case("egnn_clamped", 23, True, coord_scale=4000.0, loc_scale=6.0)    # tot_f hits the +-100 clamp (:310)
    case("egnn_norm", 24, True, coord_scale=3.0, norm=True)              # F.normalize of the Gram feature (:271-272); self loops: r = 0


def dataset_case(n_systems=6, n_balls=5, seed=43):
    """Row 8f-3: a tiny on-disk N-body dataset written by the reference's own simulator in the
    reference's file format (generate_dataset.py:84-92 -> {loc,vel,edges,charges}_<partition>_charged<name>.npy)
    and the graphs the reference's reader builds from it (datasets/nbody/dataset.py:14-113, with a
    plain-attribute stand-in for torch_geometric.data.Data).  The .npy files are committed as fixtures
    (data); the expected per-graph tensors go to dataset_nbody5.npz."""
    import pickle, tempfile
    sys.path.insert(0, os.path.join(REF, "datasets", "nbody", "datagen"))
    from system import System
    np.random.seed(seed)
    loc, vel, edges, charges, cfgs = [], [], [], [], []
    for _ in range(n_systems):
        s = System(n_isolated=n_balls, n_stick=0, n_hinge=0)
        X, V = [], []
        for t in range(4100):
            s.simulate_one_step()
            if t % 100 == 0:
                X.append(s.X.copy()); V.append(s.V.copy())
        loc.append(np.array(X)); vel.append(np.array(V)); edges.append(s.edges); charges.append(s.charges)
        cfgs.append(s.configuration())
    name = f"{n_balls}_0_0"
    fix = os.path.join(OUT, "nbody_tiny")
    os.makedirs(fix, exist_ok=True)
    tmp = tempfile.mkdtemp()
    for d in (fix, tmp):
        np.save(os.path.join(d, f"loc_train_charged{name}.npy"), np.array(loc))
        np.save(os.path.join(d, f"vel_train_charged{name}.npy"), np.array(vel))
        np.save(os.path.join(d, f"edges_train_charged{name}.npy"), np.array(edges))
        np.save(os.path.join(d, f"charges_train_charged{name}.npy"), np.array(charges))
    with open(os.path.join(tmp, f"cfg_train_charged{name}.pkl"), "wb") as f:   # the reference reader opens it
        pickle.dump(cfgs, f)

    class Data:   # stand-in: attribute bag with .to()
        def __init__(self, **kw):
            self.__dict__.update(kw)
        def to(self, *_a, **_k):
            return self
        def __repr__(self):
            return "Data(" + ", ".join(f"{k}={list(v.shape)}" for k, v in self.__dict__.items()) + ")"
    tg = types.ModuleType("torch_geometric"); tgd = types.ModuleType("torch_geometric.data")
    tgd.Data = Data; tg.data = tgd
    sys.modules["torch_geometric"] = tg; sys.modules["torch_geometric.data"] = tgd
    sys.path.insert(0, REF)
    from datasets.nbody.dataset import NBodySystemDataset
    out = {}
    for tag, rate, C, ms in (("r50", 0.5, 3, 1e8), ("r00", 0.0, 2, 4), ("r30", 0.3, 3, 1e8)):
        ds = NBodySystemDataset(name, tmp, virtual_channels=C, partition="train", max_samples=ms, frame_0=30, frame_T=40,
                                cutoff_rate=rate)
        out[f"{tag}/n"] = np.array(len(ds)); out[f"{tag}/C"] = np.array(C); out[f"{tag}/rate"] = np.array(rate)
        out[f"{tag}/max_samples"] = np.array(int(ms))
        for i in range(len(ds)):
            for k, v in ds[i].__dict__.items():
                out[f"{tag}/{i}/{k}"] = v.detach().cpu().numpy()
    np.savez_compressed(os.path.join(OUT, "dataset_nbody5.npz"), **out)
    print("dataset_nbody5:", len(out), "arrays")


def fastrf_cases():
    """Row 8f-4: the reference's FastRF class (models/FastRF.py) on small ragged batches: outputs and all
    gradients, same file layout as the FastEGNN cases."""
    _install_pyg_stand_in()
    sys.path.insert(0, REF)
    from models.FastRF import FastRF

    def case(name, seed, L=2, C=4, coord_scale=300.0, hidden=64, **flags):
        torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
        gen = torch.Generator().manual_seed(seed)
        sizes, edges = [7, 4, 9], [25, 10, 25]
        ei, batch = _rand_graph_batch(gen, sizes, edges, isolate=(2, 3))
        N = sum(sizes)
        model = FastRF(node_feat_nf=2, node_attr_nf=0, edge_attr_nf=2, hidden_nf=hidden, virtual_channels=C, n_layers=L, **flags)
        with torch.no_grad():
            for k, prm in model.named_parameters():
                if k.endswith(("coord_mlp_r.2.weight", "coord_mlp_r_virtual.2.weight", "coord_mlp_v_virtual.2.weight")):
                    prm.mul_(coord_scale)
        inp = dict(node_feat=torch.rand(N, 2, generator=gen), node_loc=torch.randn(N, 3, generator=gen) * 2.0,
                   node_vel=torch.randn(N, 3, generator=gen) * 0.5, edge_index=ei, data_batch=batch,
                   edge_attr=torch.rand(ei.size(1), 2, generator=gen))
        inp["loc_mean"] = _loc_mean(inp["node_loc"], batch, C)
        target = torch.randn(N, 3, generator=gen); wv = torch.randn(len(sizes), 3, C, generator=gen)
        leaf = {k: inp[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
        loc, vloc = model(**{**inp, **leaf})
        loss = torch.nn.functional.mse_loss(loc, target) + 0.3 * (vloc * wv).sum() / vloc.numel()
        loss.backward()
        out = {f"in/{k}": v.numpy() for k, v in inp.items()}
        out["in/target"], out["in/wv"] = target.numpy(), wv.numpy()
        for k, prm in model.named_parameters():
            out[f"p/{k}"] = prm.detach().numpy()
            out[f"gp/{k}"] = (prm.grad if prm.grad is not None else torch.zeros_like(prm)).numpy()
        for k, v in leaf.items():
            out[f"gin/{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
        out["out/loc"], out["out/vloc"] = loc.detach().numpy(), vloc.detach().numpy()
        g = flags.get("gravity")
        meta = dict(nf=2, na=0, ea=2, H=hidden, C=C, L=L, residual=1, attention=int(flags.get("attention", False)),
                    normalize=int(flags.get("normalize", False)), tanh=int(flags.get("tanh", False)),
                    has_gravity=int(g is not None), gravity=np.array(g if g is not None else [0, 0, 0], dtype=np.float32))
        for k, v in meta.items():
            out[f"meta/{k}"] = np.asarray(v)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "loss", float(loss))

    if "--fastrf-wide" in sys.argv:   # hidden_nf above 64: the sibling on the unfused wide path
        case("fastrf_h128", 24, hidden=128, attention=True, gravity=[0, -1, 0])
        return
    case("fastrf_plain", 21)
    case("fastrf_allflags", 22, attention=True, normalize=True, tanh=True, gravity=[0, -1, 0])
    case("fastrf_c16", 23, C=16, coord_scale=100.0, gravity=[0, -1, 0])


def main():
    if "--fastrf" in sys.argv or "--fastrf-wide" in sys.argv:
        fastrf_cases()
        return
    if "--dataset" in sys.argv:
        dataset_case()
        return
    if "--egnn" in sys.argv or "--egnn-wide" in sys.argv:
        egnn_cases()
        return
    FastEGNN = _import_reference()
    if "--train-only" in sys.argv:
        nb = nbody_case()
        train_case(FastEGNN, "train_nbody5", nb)
        gen = torch.Generator().manual_seed(77)
        sizes = [11, 14, 12]
        ei, batch = _rand_graph_batch(gen, sizes, [40, 50, 45])
        loc = torch.randn(sum(sizes), 3, generator=gen)
        rg = dict(node_feat=torch.rand(sum(sizes), 2, generator=gen), node_loc=loc,
                  node_vel=torch.randn(sum(sizes), 3, generator=gen) * 0.3, edge_index=ei, data_batch=batch,
                  loc_mean=_loc_mean(loc, batch, 3), edge_attr=torch.rand(ei.size(1), 2, generator=gen))
        train_case(FastEGNN, "train_ragged_simulation", rg, sigma=1.0, ragged_sizes=sizes)
        return
    if "--wide" in sys.argv:   # hidden_nf above 64 (main_nbody.py:27 --dim_hidden): the unfused wide path's golden
        run_case(FastEGNN, "wide_h128_two_graphs", sizes=[40, 23], edges=[300, 150], nf=2, na=0, ea=2, C=4, H=128, L=2, seed=31,
                 attention=True, gravity=[0, -1, 0], coord_scale=100.0, loc_scale=1.0)
        return
    if "--act" in sys.argv:   # act_fn other than the default SiLU: only these files are (re)written
        common = dict(sizes=[7, 4, 9], edges=[25, 10, 25], nf=2, na=0, ea=2, C=4, isolate=(2, 3), L=2, coord_scale=300.0,
                      gravity=[0, -1, 0])
        run_case(FastEGNN, "act_relu", seed=21, act="relu", **common)
        run_case(FastEGNN, "act_leaky_relu", seed=22, act="leaky_relu", act_param=0.2, attention=True, **common)
        run_case(FastEGNN, "act_tanh", seed=23, act="tanh", **common)
        # (sigmoid / softplus are positive at 0: the coordinate heads see 64 inputs of one sign -- a smaller head scale keeps
        # the layer-0 virtual-head gradients out of the fp32 rounding noise of two differently ordered evaluations)
        run_case(FastEGNN, "act_sigmoid", seed=24, act="sigmoid", **{**common, "coord_scale": 30.0})
        run_case(FastEGNN, "act_elu", seed=25, act="elu", act_param=1.0, tanh=True, **common)
        run_case(FastEGNN, "act_gelu", seed=26, act="gelu", **common)
        run_case(FastEGNN, "act_softplus", seed=27, act="softplus", act_param=1.5, **{**common, "coord_scale": 30.0})
        return
    # equivariant_test.py shape: 10 nodes, 20 random directed edges, nf=1, ea=1, C=3
    run_case(FastEGNN, "equiv10", sizes=[10], edges=[20], nf=1, na=0, ea=1, C=3, seed=1,
             loc_scale=5.0)
    # ragged 3-graph batch, isolated node, gravity on, trained-like coordinate heads
    common = dict(sizes=[7, 4, 9], edges=[25, 10, 25], nf=2, na=0, ea=2, C=4, isolate=(2, 3))
    run_case(FastEGNN, "ragged3_gravity", seed=2, gravity=[0, -1, 0], coord_scale=300.0, **common)
    run_case(FastEGNN, "ragged3_default_init", L=2, seed=3, **common)
    run_case(FastEGNN, "ragged3_attention", L=2, seed=4, attention=True, coord_scale=300.0, **common)
    run_case(FastEGNN, "ragged3_normalize", L=2, seed=5, normalize=True, coord_scale=300.0, **common)
    run_case(FastEGNN, "ragged3_tanh", L=2, seed=6, tanh=True, coord_scale=300.0, **common)
    run_case(FastEGNN, "ragged3_allflags", L=2, seed=7, attention=True, normalize=True, tanh=True,
             gravity=[0, -1, 0], coord_scale=300.0, **common)
    run_case(FastEGNN, "ragged3_noresidual", L=2, seed=8, residual=False, coord_scale=300.0, **common)
    run_case(FastEGNN, "ragged3_nodeattr", L=2, seed=9, coord_scale=300.0,
             **{**common, "na": 3})
    run_case(FastEGNN, "ragged3_gravity_fp64", seed=2, gravity=[0, -1, 0], coord_scale=300.0,
             dtype=torch.float64, **common)
    # C=16 / two layers: the cfg-4 channel count on a small graph (tile/loop coverage)
    run_case(FastEGNN, "c16_two_graphs", sizes=[40, 23], edges=[300, 150], nf=2, na=0, ea=2,
             C=16, L=2, seed=10, gravity=[0, -1, 0], coord_scale=100.0, loc_scale=1.0)
    # cfg-1: reference simulator frames, top-k cutoff edges, harness edge_attr augmentation
    nb = nbody_case()
    run_case(FastEGNN, "nbody5_cfg1", sizes=None, edges=None, nf=2, na=0, ea=2, C=3, seed=43,
             explicit=nb)
    run_case(FastEGNN, "nbody5_cfg1_trained", sizes=None, edges=None, nf=2, na=0, ea=2, C=3,
             seed=43, explicit=nb, coord_scale=100.0)


if __name__ == "__main__":
    main()
