"""Host-side analysis for VERDICT round 3 item 8 (edge backward below the atomic rate): how many of the col-side 256-byte atomic
rows of edge_bwd could be combined in an LDS cache of K source rows before they leave the workgroup?

The edge backward scatters one [Q | x] gradient row per edge into g_QX_src[col] with fp32 atomics (5.8 M rows per launch at cfg4,
the atomic unit's rate: ~7 G row updates / s).  A workgroup owns a contiguous run of CSR rows (E / 256 edges); a column that
recurs inside the run could be summed in LDS and flushed once.  This script builds the cfg4 frame (bench.py make_frame: 100 000
points uniform in [0, 0.965]^3, r = 0.035), orders the nodes as the file gives them (random in space) or along a Morton curve
(fastegnn_amd.sharded.morton_order), walks the edges of sample workgroups in CSR order through an LRU cache of K rows and counts
hits (= atomics saved; every miss evicts one row = one atomic).  272 bytes per cached row: K = 128 is 34 KB of LDS, K = 256 68 KB
(edge_bwd has ~9 KB free today, ~27 KB with f16x2 images)."""
import sys
from collections import OrderedDict

import numpy as np
from scipy.spatial import cKDTree


def frame(n=100000, seed=43, radius=0.035):
    rng = np.random.default_rng(seed)
    loc = rng.random((n, 3)) * 0.965
    pairs = cKDTree(loc).query_pairs(radius, output_type="ndarray")
    row = np.concatenate([pairs[:, 0], pairs[:, 1]])
    col = np.concatenate([pairs[:, 1], pairs[:, 0]])
    return loc, row, col


def morton(loc, bits=10):
    g = ((loc - loc.min(0)) / (loc.max(0) - loc.min(0)) * (2 ** bits - 1)).astype(np.int64)
    code = np.zeros(len(loc), dtype=np.int64)
    for b in range(bits):
        for k in range(3):
            code |= ((g[:, k] >> b) & 1) << (3 * b + k)
    return np.argsort(code, kind="stable")


def lru_hits(cols, K):
    cache, hits = OrderedDict(), 0
    for c in cols:
        if c in cache:
            hits += 1
            cache.move_to_end(c)
        else:
            if len(cache) >= K:
                cache.popitem(last=False)
            cache[c] = True
    return hits


def main():
    loc, row, col = frame()
    E = len(row)
    print(f"cfg4 frame: {len(loc)} nodes, {E} directed edges, mean degree {E / len(loc):.1f}")
    for name in ("file order (random in space)", "Morton order"):
        if name.startswith("Morton"):
            order = morton(loc)
            inv = np.empty_like(order)
            inv[order] = np.arange(len(order))
            r, c = inv[row], inv[col]
        else:
            r, c = row, col
        o = np.argsort(r, kind="stable")
        r, c = r[o], c[o]
        per_wg = E // 256
        rng = np.random.default_rng(1)
        wgs = rng.choice(256, size=12, replace=False)
        print(f"\n{name}: a workgroup walks {per_wg} edges = {per_wg * len(loc) // E} rows")
        distinct = np.mean([len(np.unique(c[w * per_wg:(w + 1) * per_wg])) for w in wgs])
        print(f"  distinct columns per workgroup run: {distinct:.0f}  (each cached row is 272 B: {distinct * 272 / 1024:.0f} KB to hold them all)")
        for K in (32, 64, 128, 256, 512, 1024, 4096):
            h = np.mean([lru_hits(c[w * per_wg:(w + 1) * per_wg].tolist(), K) for w in wgs]) / per_wg
            print(f"  LRU cache of {K:5d} rows ({K * 272 / 1024:6.0f} KB): hit rate {h:6.1%} -> atomics per launch {E * (1 - h) / 1e6:5.2f} M of {E / 1e6:.2f} M")


if __name__ == "__main__":
    main()
