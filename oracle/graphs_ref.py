"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the per-frame graph construction of the Water-3D dataset
(``/root/reference/datasets/simulation/dataset.py:80,96-101``).

``radius_graph`` there is torch_cluster's (through torch_geometric==2.5.2, requirements.txt:14; the wheel is
absent from this image): all ordered pairs of distinct points within distance r, no self loops.  It is
restated as a float32 brute force (exact for any N) and, for large N, through scipy's cKDTree with the
boundary re-decided in float32.  PARITY UNPINNED against torch_cluster itself (no reference test or
fixture exists for it); pinned against the brute force only."""
import numpy as np


def _d2_f32(a, b):
    d = (a - b).astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]


def radius_graph_bruteforce(loc: np.ndarray, r: float):
    loc = loc.astype(np.float32)
    d2 = _d2_f32(loc[:, None, :], loc[None, :, :])
    np.fill_diagonal(d2, np.inf)
    i, j = np.nonzero(d2 <= np.float32(r) * np.float32(r))     # row-major: grouped by i, j ascending
    return np.stack([i, j]).astype(np.int64), np.sqrt(d2[i, j]).astype(np.float32)


def radius_graph_kdtree(loc: np.ndarray, r: float):
    from scipy.spatial import cKDTree
    loc32 = loc.astype(np.float32)
    pairs = cKDTree(loc32.astype(np.float64)).query_pairs(r * (1 + 1e-5), output_type="ndarray")
    i = np.concatenate([pairs[:, 0], pairs[:, 1]]); j = np.concatenate([pairs[:, 1], pairs[:, 0]])
    d2 = _d2_f32(loc32[i], loc32[j])
    keep = d2 <= np.float32(r) * np.float32(r)
    i, j, d2 = i[keep], j[keep], d2[keep]
    order = np.lexsort((j, i))
    return np.stack([i[order], j[order]]).astype(np.int64), np.sqrt(d2[order]).astype(np.float32)


def cutoff_edges(edge_index: np.ndarray, dist: np.ndarray, cutoff_rate: float):
    """cutoff_edge (:96-101): ascending sort by length, first int(E*(1-rate)); stable on ties."""
    keep = int(edge_index.shape[1] * (1 - cutoff_rate))
    order = np.argsort(dist, kind="stable")[:keep]
    return edge_index[:, order], dist[order]
