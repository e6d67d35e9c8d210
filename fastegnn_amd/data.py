"""Batch format and dataset readers: what sits between the on-disk data and
``FastEGNN.forward`` in the reference's training harness.

* ``Frame``  -- one graph with the fields the reference datasets put into a ``torch_geometric.data.Data``
  (``datasets/nbody/dataset.py:97-98``, ``datasets/simulation/dataset.py:92-93``).
* ``collate`` / ``Batch`` -- the mini-batch the harness reads with ``data['...']``
  (``utils/train.py:36-53``): node- and edge-level tensors concatenated, ``edge_index`` shifted by the
  cumulative node count, ``batch`` [N], ``ptr`` [B+1], ``loc_mean`` stacked to [B,3,C].  This is the
  documented behaviour of PyG 2.5.2's collate for these fields (PyG is absent here: parity unpinned).
* ``NBodySystemDataset`` -- reader for the ``.npy`` files of ``datasets/nbody/datagen``
  (``datasets/nbody/dataset.py:44-113``), pinned by tests/golden/dataset_nbody5.npz.
* ``Simulation`` -- reader for the Water-3D ``.h5`` files (``datasets/simulation/dataset.py:46-101``);
  needs ``h5py`` and builds the radius graph on the GPU through the C-ABI (graphs.py).
* ``DataLoader`` -- ``batch_size`` / ``shuffle`` / ``drop_last`` iteration over frames (``main_nbody.py:93-96``).

All processing is vectorised over the systems of a file and runs on the device the caller names.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, fields
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

_NODE_KEYS = ("loc_0", "loc_t", "vel_0", "node_feat", "node_attr")


@dataclass
class Frame:
    edge_index: torch.Tensor   # int64 [2,E]
    edge_attr: torch.Tensor    # [E,1] edge length at frame 0
    loc_0: torch.Tensor        # [n,3]
    loc_t: torch.Tensor        # [n,3]
    vel_0: torch.Tensor        # [n,3]
    node_feat: torch.Tensor    # [n,2]  (|vel|, charge or type / max)
    node_attr: torch.Tensor    # [n,1]
    loc_mean: torch.Tensor     # [1,3,C]

    @property
    def num_nodes(self) -> int:
        return self.node_feat.size(0)

    def to(self, device) -> "Frame":
        return Frame(**{f.name: getattr(self, f.name).to(device) for f in fields(self)})

    def __getitem__(self, key: str) -> torch.Tensor:
        return getattr(self, key)


class Batch:
    """Mini-batch with the mapping interface the harness uses (``data['ptr']``, ``data.to(device)``,
    ``data.detach()``)."""

    def __init__(self, **tensors: torch.Tensor):
        self._t = dict(tensors)

    def __getitem__(self, key: str) -> torch.Tensor:
        return self._t[key]

    def __getattr__(self, key: str) -> torch.Tensor:
        try:
            return self.__dict__["_t"][key]
        except KeyError:
            raise AttributeError(key) from None

    def get(self, key: str, default=None):
        return self._t.get(key, default)

    def keys(self):
        return self._t.keys()

    @property
    def num_graphs(self) -> int:
        return self._t["ptr"].numel() - 1

    @property
    def num_nodes(self) -> int:
        return self._t["batch"].numel()

    def to(self, device, non_blocking: bool = False) -> "Batch":
        return Batch(**{k: v.to(device, non_blocking=non_blocking) for k, v in self._t.items()})

    def detach(self) -> "Batch":
        return Batch(**{k: v.detach() for k, v in self._t.items()})

    def pin_memory(self) -> "Batch":
        return Batch(**{k: v.pin_memory() for k, v in self._t.items()})

    def __repr__(self) -> str:
        return "Batch(" + ", ".join(f"{k}={list(v.shape)}" for k, v in self._t.items()) + ")"


def collate(frames: Sequence[Frame]) -> Batch:
    if len(frames) == 0:
        raise ValueError("collate: empty list of frames")
    dev = frames[0].node_feat.device
    counts = torch.tensor([f.num_nodes for f in frames], dtype=torch.int64)
    ptr = torch.zeros(len(frames) + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(counts, 0)
    out = {k: torch.cat([getattr(f, k) for f in frames], 0) for k in _NODE_KEYS}
    out["edge_attr"] = torch.cat([f.edge_attr for f in frames], 0)
    out["edge_index"] = torch.cat([f.edge_index + int(o) for f, o in zip(frames, ptr[:-1])], 1)
    out["loc_mean"] = torch.cat([f.loc_mean for f in frames], 0)
    out["batch"] = torch.repeat_interleave(torch.arange(len(frames), dtype=torch.int64), counts).to(dev)
    out["ptr"] = ptr.to(dev)
    return Batch(**out)


def _node_feat(vel_0: torch.Tensor, kind: torch.Tensor) -> torch.Tensor:
    # [|vel|, kind / max(kind)]   (datasets/nbody/dataset.py:89-92, datasets/simulation/dataset.py:84-87)
    return torch.cat([vel_0.pow(2).sum(-1, keepdim=True).sqrt(), kind / kind.max()], -1)


def _loc_mean(loc_0: torch.Tensor, C: int) -> torch.Tensor:
    return loc_0.mean(0).unsqueeze(-1).repeat(1, C).unsqueeze(0)


class NBodySystemDataset:
    """Frames (frame_0 -> frame_T) of the charged N-body systems written by ``datasets/nbody/datagen``.

    Same constructor arguments as ``datasets/nbody/dataset.py:18``; ``rotation`` replaces the
    reference's unseeded ``random_rotate()`` of the test partition (``:77-83``): pass a [3,3] matrix, a
    callable ``i -> [3,3]`` or None.  Edges: the ``int(n(n-1)(1-cutoff_rate))`` shortest ordered pairs of the
    complete graph in ascending length (``:102-113``); the two directions of a pair have the same
    length, so when that count is odd the direction kept for the last pair is unspecified, as in the
    reference (``torch.topk`` tie order)."""

    def __init__(self, dataset_name, data_dir, virtual_channels, partition="train", max_samples=1e8, frame_0=30,
                 frame_T=40, cutoff_rate=0.0, device="cpu", rotation=None):
        self.partition, self.virtual_channels, self.cutoff_rate = partition, int(virtual_channels), float(cutoff_rate)
        suffix = f"{partition}_charged{dataset_name}"
        n_keep = int(max_samples)
        # .npy layout (generate_dataset.py:84-92): loc/vel float64 [S,T,n,3], charges [S,n,1], edges [S,n,n]
        loc = np.load(os.path.join(data_dir, f"loc_{suffix}.npy"), mmap_mode="r")[:n_keep]
        vel = np.load(os.path.join(data_dir, f"vel_{suffix}.npy"), mmap_mode="r")[:n_keep]
        charges = np.load(os.path.join(data_dir, f"charges_{suffix}.npy"))[:n_keep]
        if loc.ndim != 4 or loc.shape[-1] != 3 or vel.shape != loc.shape:
            raise ValueError(f"NBodySystemDataset: unexpected array shapes {loc.shape} / {vel.shape}")
        self.num_node_r = loc.shape[-2]
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(device)  # noqa: E731
        self.data = self._process(f32(loc[:, frame_0]), f32(vel[:, frame_0]), f32(loc[:, frame_T]), f32(charges), rotation)

    def _process(self, loc_0, vel_0, loc_t, charges, rotation) -> List[Frame]:
        S, n, _ = loc_0.shape
        if rotation is not None:
            R = torch.stack([torch.as_tensor(rotation(i) if callable(rotation) else rotation) for i in range(S)])
            R = R.to(loc_0.device, torch.float32)
            loc_0, loc_t, vel_0 = loc_0 @ R, loc_t @ R, vel_0 @ R
        k = int(n * (n - 1) * (1 - self.cutoff_rate))
        if loc_0.is_cuda and n <= 128:
            from .graphs import nbody_cutoff_edges
            # device tensors: one workgroup per system sorts its n(n-1) pairs in LDS (csrc/graphs.hip)
            ei, ea = nbody_cutoff_edges(loc_0, k)
            ea = ea.unsqueeze(-1)                                                              # [S,k,1]
        else:   # host-side dataset build (what the reference does), or systems beyond the kernel's 128 particles
            dist = torch.cdist(loc_0, loc_0, p=2) + torch.eye(n, device=loc_0.device) * 1e18
            idc = torch.topk(dist.reshape(S, n * n), k, dim=1, largest=False).indices
            ei = torch.stack([idc.div(n, rounding_mode="trunc"), idc.remainder(n)], 1).long()  # [S,2,k]
            d = loc_0.gather(1, ei[:, 0, :, None].expand(-1, -1, 3)) - loc_0.gather(1, ei[:, 1, :, None].expand(-1, -1, 3))
            ea = d.pow(2).sum(-1).sqrt().unsqueeze(-1)                                        # [S,k,1]
        return [Frame(ei[s], ea[s], loc_0[s], loc_t[s], vel_0[s], _node_feat(vel_0[s], charges[s]), charges[s],
                      _loc_mean(loc_0[s], self.virtual_channels)) for s in range(S)]

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, i: int) -> Frame:
        return self.data[i]


def water3d_frame(loc_0, vel_0, loc_t, node_type, virtual_channels, radius=0.035, cutoff_rate=0.0, rotation=None) -> Frame:
    """``get_graph_step`` (``datasets/simulation/dataset.py:71-93``) for one frame of device tensors: radius
    graph without self loops, shortest-fraction cutoff, edge length attribute, node features, loc_mean."""
    from .graphs import cutoff_edges, radius_graph
    if rotation is not None:
        R = torch.as_tensor(rotation() if callable(rotation) else rotation).to(loc_0.device, torch.float32)
        loc_0, loc_t, vel_0 = loc_0 @ R, loc_t @ R, vel_0 @ R
    ei, dist = radius_graph(loc_0, radius)
    ei, dist = cutoff_edges(ei, dist, cutoff_rate)
    return Frame(ei, dist.unsqueeze(-1), loc_0, loc_t, vel_0, _node_feat(vel_0, node_type), node_type,
                 _loc_mean(loc_0, int(virtual_channels)))


def read_trajectories(directory: str, partition: str):
    """Yields ``(key, position [T,n,3], particle_type [n])`` of every trajectory of a partition, in file order.

    Two containers with the same fields: the reference's ``<partition>.h5`` (one HDF5 group per trajectory holding
    ``position`` and ``particle_type``, ``datasets/simulation/dataset.py:46-56``; needs ``h5py``) and
    ``<partition>.npz`` with arrays named ``<key>/position`` and ``<key>/particle_type`` (what
    ``numpy.savez`` of the same groups gives; readable without h5py).  The ``.h5`` file wins when both exist and
    h5py is importable; a lone ``.h5`` without h5py raises."""
    h5, npz = os.path.join(directory, f"{partition}.h5"), os.path.join(directory, f"{partition}.npz")
    if os.path.exists(h5):
        try:
            import h5py
        except ImportError as e:
            if not os.path.exists(npz):
                raise ImportError(f"fastegnn_amd.data: {h5} needs h5py (absent); convert it to {npz} "
                                  "(arrays '<key>/position', '<key>/particle_type')") from e
        else:
            with h5py.File(h5, "r") as f:
                for key in list(f.keys()):
                    yield key, np.array(f[key]["position"]), np.array(f[key]["particle_type"])
            return
    if not os.path.exists(npz):
        raise FileNotFoundError(f"fastegnn_amd.data: neither {h5} nor {npz} exists")
    with np.load(npz) as z:
        keys = []
        for name in z.files:                      # file order, first appearance of each trajectory key
            k = name.rsplit("/", 1)[0]
            if k not in keys:
                keys.append(k)
        for k in keys:
            yield k, z[f"{k}/position"], z[f"{k}/particle_type"]


class Simulation:
    """Water-3D style particle trajectories (``<data_dir>/<dataset_name>/<partition>.h5`` -- or ``.npz``, see
    ``read_trajectories``: one group per trajectory with ``position`` [T,n,3] and ``particle_type`` [n]).

    Same constructor arguments as ``datasets/simulation/dataset.py:17``.  The reference draws 15 start
    frames per trajectory with the unseeded ``random.randint(0, 250)`` (``:58``) and shuffles the frames
    (``:31``); here both come from ``seed``.  Per frame: ``vel_0 = pos[f+1] - pos[f]``, target
    ``pos[f + delta_t]``, edges = radius graph (r = 0.035, no self loops) reduced to the shortest
    ``1 - cutoff_rate`` fraction, built on the GPU (graphs.radius_graph / graphs.cutoff_edges)."""

    RADIUS = 0.035
    FRAMES_PER_TRAJECTORY = 15

    def __init__(self, dataset_name, data_dir, virtual_channels, partition="train", max_samples=1e8, delta_t=15,
                 cutoff_rate=0.0, device="cuda", rotation=None, seed=0, radius=None):
        self.virtual_channels, self.cutoff_rate, self.delta_t = int(virtual_channels), float(cutoff_rate), int(delta_t)
        self.radius = self.RADIUS if radius is None else float(radius)
        rng = np.random.default_rng(seed)
        self.data: List[Frame] = []
        self.frames: List[tuple] = []      # (trajectory key, start frame) of every sample, in the order drawn
        max_samples = int(max_samples)
        for key, position, particle_type in read_trajectories(os.path.join(data_dir, dataset_name), partition):
            ptype = torch.from_numpy(np.asarray(particle_type)).float().reshape(-1, 1)
            pos = torch.from_numpy(np.asarray(position)).float()
            n_frames = min(self.FRAMES_PER_TRAJECTORY, max_samples - len(self.data))
            last = min(250, pos.size(0) - 1 - max(1, self.delta_t))
            for fr in rng.integers(0, last + 1, size=max(n_frames, 0)):
                fr = int(fr)
                self.frames.append((key, fr))
                self.data.append(water3d_frame(pos[fr].to(device), (pos[fr + 1] - pos[fr]).to(device),
                                               pos[fr + self.delta_t].to(device), ptype.to(device),
                                               self.virtual_channels, self.radius, self.cutoff_rate, rotation))
            if len(self.data) >= max_samples:
                break
        order = rng.permutation(len(self.data))
        self.data = [self.data[i] for i in order]
        self.frames = [self.frames[i] for i in order]

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, i: int) -> Frame:
        return self.data[i]


class DataLoader:
    """``batch_size`` / ``shuffle`` / ``drop_last`` iteration yielding ``Batch`` objects
    (the use of ``torch_geometric.loader.DataLoader`` at ``main_nbody.py:93-96``)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, generator: Optional[torch.Generator] = None,
                 device=None):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.generator, self.device = generator, device

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def __iter__(self) -> Iterator[Batch]:
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for i in range(len(self)):
            b = collate([self.dataset[j] for j in order[i * self.batch_size:(i + 1) * self.batch_size]])
            yield b.to(self.device) if self.device is not None else b
