"""Graph construction on the GPU: what the Water-3D dataset does per frame on
the host (``datasets/simulation/dataset.py:80,96-101``) -- ``radius_graph(r)`` without self loops and the
"keep the shortest (1 - cutoff_rate) fraction" cutoff -- as C-ABI calls of ``libfastegnn_hip.so``."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from . import _lib as K
from .model import _stream


def radius_graph(loc: torch.Tensor, r: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ordered pairs (i, j != i) with ||loc_i - loc_j|| <= r (decided on the fp32 squared distance).
    Returns (edge_index int64 [2,E] grouped by edge_index[0] with ascending edge_index[1], dist fp32 [E])."""
    assert loc.is_cuda and loc.dim() == 2 and loc.size(1) == 3
    loc = loc.contiguous().float()
    N, dev = loc.size(0), loc.device
    L = K.lib()
    nbytes = L.fastegnn_radius_graph_ws_bytes(N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    E = C.c_int64(0)
    st = _stream(dev)
    K.check(L.fastegnn_radius_graph_count(K.ptr(loc), N, float(r), K.ptr(ws), nbytes, C.byref(E), st),
            "fastegnn_radius_graph_count")
    ei = torch.empty(2, E.value, dtype=torch.int64, device=dev)
    dist = torch.empty(E.value, dtype=torch.float32, device=dev)
    K.check(L.fastegnn_radius_graph_fill(K.ptr(loc), N, float(r), K.ptr(ws), nbytes, E.value, K.ptr(ei), K.ptr(dist),
                                         st), "fastegnn_radius_graph_fill")
    return ei, dist


def cutoff_edges(edge_index: torch.Tensor, dist: torch.Tensor, cutoff_rate: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Keep the ``int(E * (1 - cutoff_rate))`` shortest edges, in ascending length (cutoff_edge of the
    reference datasets); equal lengths keep their input order."""
    E = edge_index.size(1)
    keep = int(E * (1 - cutoff_rate))
    dev = edge_index.device
    out = torch.empty(2, keep, dtype=torch.int64, device=dev)
    dout = torch.empty(keep, dtype=torch.float32, device=dev)
    if keep:
        L = K.lib()
        nbytes = L.fastegnn_cutoff_tmp_bytes(E)
        tmp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        K.check(L.fastegnn_cutoff_edges(K.ptr(edge_index.contiguous()), K.ptr(dist.contiguous()), E, keep, K.ptr(out),
                                        K.ptr(dout), K.ptr(tmp), nbytes, _stream(dev)), "fastegnn_cutoff_edges")
    return out, dout


NBODY_MAX_PARTICLES = 128


def nbody_cutoff_edges(loc: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """For each system of ``loc`` [S,n,3] (n <= 128) the ``k`` shortest ordered pairs of the complete graph without self
    loops, in ascending length (``datasets/nbody/dataset.py:102-113``: cdist + cutoff_edge on the host).  Returns
    (edge_index int64 [S,2,k] with ids local to the system, dist fp32 [S,k]); equal lengths come in ascending i*n+j."""
    assert loc.is_cuda and loc.dim() == 3 and loc.size(2) == 3
    loc = loc.contiguous().float()
    S, n, dev = loc.size(0), loc.size(1), loc.device
    ei = torch.empty(S, 2, k, dtype=torch.int64, device=dev)
    dist = torch.empty(S, k, dtype=torch.float32, device=dev)
    K.check(K.lib().fastegnn_nbody_cutoff_edges(K.ptr(loc), S, n, k, K.ptr(ei), K.ptr(dist), _stream(dev)),
            "fastegnn_nbody_cutoff_edges")
    return ei, dist
