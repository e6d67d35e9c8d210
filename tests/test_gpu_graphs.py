"""-m gpu: device-side graph construction (fastegnn_amd/graphs.py) against the CPU restatement
(oracle/graphs_ref.py): exact edge lists (integer parity) for the radius graph and the cutoff."""
import numpy as np
import pytest
import torch

from fastegnn_amd.graphs import cutoff_edges, radius_graph
from oracle import graphs_ref as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,r,scale", [(1, 0.1, 1.0), (2, 0.5, 0.1), (300, 0.12, 1.0), (1500, 0.035, 0.3), (700, 5.0, 1.0)])
def test_radius_graph_equals_bruteforce(N, r, scale):
    g = np.random.RandomState(N)
    loc = (g.rand(N, 3) * scale - 0.3 * scale).astype(np.float32)      # negative coordinates too
    if N >= 300:
        loc[7] = loc[3]                                               # coincident points (distance 0)
    ei_ref, d_ref = G.radius_graph_bruteforce(loc, r)
    ei, d = radius_graph(torch.from_numpy(loc).cuda(), r)
    assert ei.shape[1] == ei_ref.shape[1]
    assert np.array_equal(ei.cpu().numpy(), ei_ref)
    assert np.allclose(d.cpu().numpy(), d_ref, rtol=1e-6, atol=1e-9)


def test_radius_graph_water3d_size_equals_kdtree_reference():
    # BASELINE configs[3] frame: 100k points, r = 0.035 -> ~1.9 M directed edges
    g = torch.Generator().manual_seed(43)
    loc = torch.rand(100000, 3, generator=g) * 0.965
    ei_ref, d_ref = G.radius_graph_kdtree(loc.numpy(), 0.035)
    ei, d = radius_graph(loc.cuda(), 0.035)
    assert np.array_equal(ei.cpu().numpy(), ei_ref)
    # symmetric, no self loops
    a = ei.cpu().numpy()
    assert (a[0] != a[1]).all()
    key = a[0] * 100000 + a[1]
    assert np.array_equal(np.sort(key), np.sort(a[1] * 100000 + a[0]))      # every (i,j) has its (j,i)


def test_radius_graph_small_radius_grid_bound():
    """Extent >= 256 r on every axis: the cell grid is capped at 256 cells per axis (2^24 cells, the size of the
    cell-start table); a near-cubic cloud with r < extent/256 used to ask for 257^3 cells."""
    g = torch.Generator().manual_seed(17)
    loc = torch.rand(60000, 3, generator=g)
    loc[0], loc[1] = torch.tensor([0.0, 0.0, 0.0]), torch.tensor([1.0, 1.0, 1.0])    # pin the bounding box
    loc[2] = torch.tensor([1.0, 1.0, 1.0]) - 0.001                                   # a pair in the last cell
    for r in (0.003, 0.0039):
        ei_ref, d_ref = G.radius_graph_kdtree(loc.numpy(), r)
        ei, d = radius_graph(loc.cuda(), r)
        assert ei_ref.shape[1] > 0 and np.array_equal(ei.cpu().numpy(), ei_ref)


@pytest.mark.parametrize("rate", [0.0, 0.5, 0.9, 1.0])
def test_cutoff_keeps_shortest_fraction(rate):
    g = np.random.RandomState(5)
    loc = g.rand(800, 3).astype(np.float32)
    ei_ref, d_ref = G.radius_graph_bruteforce(loc, 0.15)
    k_ref, kd_ref = G.cutoff_edges(ei_ref, d_ref, rate)
    ei, d = radius_graph(torch.from_numpy(loc).cuda(), 0.15)
    k, kd = cutoff_edges(ei, d, rate)
    assert np.array_equal(k.cpu().numpy(), k_ref)          # stable ties: identical order
    assert np.allclose(kd.cpu().numpy(), kd_ref, rtol=3e-7, atol=0)      # sqrt may differ by an ulp
    assert bool((kd[1:] >= kd[:-1]).all())


def test_device_built_graph_feeds_the_model():
    import fastegnn_amd
    g = torch.Generator().manual_seed(2)
    loc = torch.rand(5000, 3, generator=g) * 0.4
    ei, d = radius_graph(loc.cuda(), 0.035)
    ei, d = cutoff_edges(ei, d, 0.5)
    m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 3, device="cuda", gravity=[0, -1, 0])
    N = loc.size(0)
    out = m(node_feat=torch.rand(N, 2).cuda(), node_loc=loc.cuda(), node_vel=torch.zeros(N, 3).cuda(), edge_index=ei,
            data_batch=torch.zeros(N, dtype=torch.long).cuda(), loc_mean=loc.mean(0).view(1, 3, 1).repeat(1, 1, 3).cuda(),
            edge_attr=torch.stack([d, d], 1))
    assert torch.isfinite(out[0]).all()


def test_water3d_frames_collate_and_train_on_device():
    """Row 8f-3 on the GPU: positions -> water3d_frame (device radius graph + cutoff) -> collate -> one
    training step, i.e. the loop body of utils/train.py:30-179 with every tensor resident on the device."""
    import fastegnn_amd
    from fastegnn_amd import data as D
    from fastegnn_amd.train import FusedAdam, train_step
    g = torch.Generator().manual_seed(3)
    frames = []
    for n in (1500, 2200):
        pos = (torch.rand(n, 3, generator=g) * 0.25).cuda()
        vel = (torch.randn(n, 3, generator=g) * 0.003).cuda()
        kind = torch.randint(1, 4, (n, 1), generator=g).float().cuda()
        f = D.water3d_frame(pos, vel, pos + 15 * vel, kind, virtual_channels=4, radius=0.035, cutoff_rate=0.5)
        # the frame equals the host restatement of get_graph_step
        ref = G.cutoff_edges(*G.radius_graph_bruteforce(pos.cpu().numpy(), 0.035), 0.5)
        assert sorted(zip(*f.edge_index.cpu().tolist())) == sorted(zip(*ref[0].tolist()))
        torch.testing.assert_close(f.node_feat[:, 0].cpu(), vel.cpu().norm(dim=1))
        torch.testing.assert_close(f.node_feat[:, 1], kind[:, 0] / kind.max())
        assert f.loc_mean.shape == (1, 3, 4)
        frames.append(f)
    b = D.collate(frames)
    assert b["ptr"].tolist() == [0, 1500, 3700] and b["edge_index"].is_cuda
    assert torch.equal(b["batch"][b["edge_index"][0]], b["batch"][b["edge_index"][1]])
    m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 4, device="cuda", n_layers=2, gravity=[0, -1, 0])
    opt = FusedAdam(m.parameters(), lr=5e-4, weight_decay=1e-12)
    sample = torch.stack([torch.randperm(1500, generator=g)[:12], 1500 + torch.randperm(2200, generator=g)[:12]]).cuda()
    l0, mse0 = train_step(m, opt, b, sample_nodes=sample, sigma=1.0, weight=0.01)
    l1, mse1 = train_step(m, opt, b, sample_nodes=sample, sigma=1.0, weight=0.01)
    assert np.isfinite(float(l0)) and np.isfinite(float(l1)) and float(mse1) != float(mse0)


@pytest.mark.parametrize("n,rate", [(5, 0.0), (5, 0.5), (100, 0.0), (100, 0.3), (128, 0.9), (37, 0.123)])
def test_nbody_cutoff_edges_vs_topk(n, rate):
    """The k shortest ordered pairs per system (datasets/nbody/dataset.py:102-113) against torch.topk on the same
    fp32 distances: ascending lengths, no self loops, the same edge set (both directions of a pair have the same
    length, so the last edge of an odd k may be either direction -- compared as unordered pairs there)."""
    from fastegnn_amd.graphs import nbody_cutoff_edges
    g = torch.Generator().manual_seed(n)
    S = 7
    loc = torch.randn(S, n, 3, generator=g).cuda()
    k = int(n * (n - 1) * (1 - rate))
    ei, dist = nbody_cutoff_edges(loc, k)
    assert ei.shape == (S, 2, k) and dist.shape == (S, k)
    assert (ei[:, 0] != ei[:, 1]).all() and ei.min() >= 0 and ei.max() < n
    assert (dist[:, 1:] >= dist[:, :-1]).all()
    d = loc.gather(1, ei[:, 0, :, None].expand(-1, -1, 3)) - loc.gather(1, ei[:, 1, :, None].expand(-1, -1, 3))
    assert torch.allclose(d.pow(2).sum(-1).sqrt(), dist, rtol=1e-6, atol=0)
    full = (loc[:, :, None, :] - loc[:, None, :, :]).pow(2).sum(-1).sqrt() + torch.eye(n, device="cuda") * 1e18
    ref = torch.topk(full.reshape(S, n * n), k, dim=1, largest=False)
    assert torch.allclose(ref.values, dist, rtol=1e-6, atol=0)
    for s in range(S):
        got = set((ei[s, 0] * n + ei[s, 1]).tolist())
        assert len(got) == k
        want = set(ref.indices[s].tolist())
        unordered = lambda q: {(min(e // n, e % n), max(e // n, e % n)) for e in q}   # noqa: E731
        assert len(got ^ want) <= 2 and unordered(got) == unordered(want)


def test_nbody_dataset_on_device_matches_host_build(tmp_path):
    """NBodySystemDataset built from device tensors (kernel) and from host tensors (torch path) give the same frames."""
    import numpy as np
    from fastegnn_amd.data import NBodySystemDataset
    g = np.random.RandomState(0)
    S, T, n = 6, 3, 5
    for part in ("train",):
        np.save(tmp_path / f"loc_{part}_charged5.npy", g.randn(S, T, n, 3))
        np.save(tmp_path / f"vel_{part}_charged5.npy", g.randn(S, T, n, 3))
        np.save(tmp_path / f"charges_{part}_charged5.npy", g.choice([-1.0, 1.0], size=(S, n, 1)))
    kw = dict(dataset_name="5", data_dir=str(tmp_path), virtual_channels=3, partition="train", frame_0=0, frame_T=2,
              cutoff_rate=0.4)
    a = NBodySystemDataset(device="cuda", **kw)
    b = NBodySystemDataset(device="cpu", **kw)
    assert len(a) == len(b) == S
    for fa, fb in zip(a.data, b.data):
        ea_, eb_ = fa.edge_index.cpu(), fb.edge_index
        assert ea_.shape == eb_.shape
        assert set(map(tuple, ea_.t().tolist())) == set(map(tuple, eb_.t().tolist()))   # k = 12: even, whole pairs
        assert torch.allclose(fa.edge_attr.cpu(), fb.edge_attr, rtol=1e-6)
        assert torch.equal(fa.loc_0.cpu(), fb.loc_0)
