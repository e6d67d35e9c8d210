"""TEST INFRASTRUCTURE: a CPU implementation of the stage backend interface of fastegnn_amd/sharded.py,
built on the oracle's stage functions (oracle/factored.py).  It lets the world_size-2 gloo test run
the *orchestration* of the sharded path (partitioning, collectives, gradient bookkeeping) without a
GPU.  Buffer names and semantics follow include/fastegnn_hip.h."""
from __future__ import annotations

import torch

from oracle import factored as F
from oracle.fastegnn_ref import Config

H = 64


class CpuGraph:
    def __init__(self, edge_index, n_rows, n_src, row_begin):
        self.n_rows, self.n_src, self.E = n_rows, n_src, edge_index.size(1)
        row = edge_index[0] - row_begin
        self.perm = torch.sort(row, stable=True).indices
        self.row, self.col = row[self.perm], edge_index[1][self.perm]
        deg = torch.bincount(self.row, minlength=n_rows)
        self.csr = F.Csr(n_rows, None, self.row, self.col, self.perm, None, None, 1.0 / deg.clamp(min=1).float())

    def permute(self, ea):
        return None if ea is None or ea.size(1) == 0 else ea[self.perm].contiguous()


class CpuOracleBackend:
    def __init__(self, names, n_layers, cfg: Config):
        self.names, self.cfg, self.n_layers = names, cfg, n_layers

    # ---- memory ----
    def empty(self, *shape):
        return torch.zeros(*shape)

    zeros = empty

    def carve(self, shapes):
        return {k: torch.zeros(*s) for k, s in shapes.items()}

    def wpack_floats(self, Cn):
        return 4

    def wg_slab_floats(self):
        return 4

    def wg_edge_floats(self, E):
        return 4

    def wg_virt_floats(self, N, Cn, flags=0):
        return 4

    def pack_all(self, spec, N, B, graph, layer_params, wpacks):   # the oracle's stages read the parameters directly
        pass

    def wgrad_open(self, spec, N, B, graph, t, params):   # the oracle's stages contract their own weight gradients
        return None

    def wgrad_close(self, handle):
        pass

    def wg_node_floats(self, N, B, Cn):
        return 4

    # ---- prologue / epilogue ----
    def build_graph(self, ei, n_rows, n_src, row_begin, csc=True):
        return CpuGraph(ei, n_rows, n_src, row_begin)

    def pad_params(self, names, h, C_, rf, params):
        from tests.helpers import pad_reference
        return [pad_reference(n, p, h, C_, rf) for n, p in zip(names, params)]

    def build_batch(self, data_batch, N, B):
        return data_batch.clone(), F.graph_ptr(data_batch, B)

    def embed_forward(self, node_feat, nf, W, b, h):
        h.copy_(node_feat @ W.T + b)

    def embed_backward(self, node_feat, g_h, nf, W, gW, gb, g_nf):
        gW += g_h.T @ node_feat
        gb += g_h.sum(0)
        if g_nf is not None:
            g_nf.copy_(g_h @ W)

    def virtual_init(self, vnf, B, Cn, HvT):
        HvT.copy_(vnf[0].T.unsqueeze(0).expand(B, -1, -1))

    def virtual_init_backward(self, g_HvT, B, Cn, g_vnf):
        g_vnf += g_HvT.sum(0).T.unsqueeze(0)

    # ---- stages ----
    def _w(self, params):
        # params: the layer's 37-slot list -> dict with layer-local names -> LayerW
        from fastegnn_amd._lib import PARAM_SLOTS
        p = {f"L.{s}": t for s, t in zip(PARAM_SLOTS, params) if t is not None}
        return F.LayerW(p, "L", self.cfg), p

    def _G(self, grads, params):
        """gradient accumulators by layer-local name; a slot the caller left empty (the per-graph stages on ranks other
        than 0) gets a scratch tensor that is dropped afterwards"""
        from fastegnn_amd._lib import PARAM_SLOTS
        return {f"L.{s}": (g if g is not None else torch.zeros_like(p))
                for s, g, p in zip(PARAM_SLOTS, grads, params) if p is not None}

    def stage(self, name, spec, N, B, graph, t, params, grads=None, flags=0):
        from fastegnn_amd._lib import F_GQX_ACCUM
        cfg = self.cfg
        w, _ = self._w(params)
        G = self._G(grads, params) if grads is not None else None
        grav = torch.tensor(list(cfg.gravity)) if cfg.gravity is not None else None
        cnt = t["xsum"][:, 3].clamp(min=1) if "xsum" in t else None
        batch = t["batch"]
        if name == "pack_weights":
            return
        if name == "node_pre_forward":
            P, Q, A, svel, sgrav = F.node_pre_fwd(w, cfg, t["h"])
            t["P"].copy_(P); t["A"].copy_(A); t["svel"].copy_(svel)
            if sgrav is not None:
                t["sgrav"].copy_(sgrav)
            t["QX"][:N, :H] = Q
            t["QX"][:N, H:H + 3] = t["x"]
        elif name == "graph_xsum":
            t["xsum"].zero_()
            t["xsum"][:, :3].index_add_(0, batch, t["x"])
            t["xsum"][:, 3].index_add_(0, batch, torch.ones(N))
        elif name == "graph_pre_forward":
            t["Bc"].copy_(self._graph_pre(w, t, cnt)[2])
        elif name == "edge_forward":
            src = t["QX_src"]
            aggm, aggx = F.edge_fwd(w, cfg, graph.csr, t["P"], src[:, :H], t["x"], t["ea_sorted"], x_src=src[:, H:H + 3])
            t["aggm"].copy_(aggm); t["aggx"].copy_(aggx)
        elif name == "virt_forward":
            h_new, x_new, poolV, poolX = F.virt_fwd(w, cfg, t["h"], t["A"], t["Bc"], t["x"], t["vel"], t["Z"], batch,
                                                    t["aggm"], t["aggx"], t["svel"], t["sgrav"] if grav is not None else None,
                                                    grav, t.get("node_attr"))
            t["h_out"].copy_(h_new); t["x_out"].copy_(x_new); t["poolV"].copy_(poolV); t["poolX"].copy_(poolX)
        elif name == "graph_post_forward":
            inv = (1.0 / cnt).view(-1, 1, 1)
            t["Z_out"].copy_(t["Z"] + t["poolX"] * inv)
            z5 = t["HvT"] @ w.W5a.T + (t["poolV"] * inv) @ w.W5b.T + w.b5
            out = F.silu(z5) @ w.W6.T + w.b6
            t["HvT_out"].copy_(t["HvT"] + out if cfg.residual else out)
        elif name == "graph_post_backward":
            inv = (1.0 / cnt).view(-1, 1, 1)
            pm = t["poolV"] * inv
            z5 = t["HvT"] @ w.W5a.T + pm @ w.W5b.T + w.b5
            u = F.silu(z5)
            g_out = t["g_HvT_out"]
            G["L.node_mlp_virtual.2.bias"] += g_out.sum((0, 1))
            G["L.node_mlp_virtual.2.weight"] += torch.einsum("bco,bch->oh", g_out, u)
            g_z5 = (g_out @ w.W6) * F.dsilu(z5)
            G["L.node_mlp_virtual.0.bias"] += g_z5.sum((0, 1))
            G["L.node_mlp_virtual.0.weight"][:, :H] += torch.einsum("bco,bch->oh", g_z5, t["HvT"])
            G["L.node_mlp_virtual.0.weight"][:, H:] += torch.einsum("bco,bch->oh", g_z5, pm)
            t["g_HvT"].copy_((g_out if cfg.residual else 0) + g_z5 @ w.W5a)
            t["g_poolV"].copy_((g_z5 @ w.W5b) * inv)
            t["g_poolX"].copy_(t["g_Z_out"] * inv)
            t["g_Z"].copy_(t["g_Z_out"])
        elif name == "virt_backward":
            r = F.virt_bwd(w, cfg, G, "L", t["h"], t["A"], t["Bc"], t["x"], t["vel"], t["Z"], batch, t["aggm"], grav,
                           t["g_h_out"], t["g_x_out"], t["g_poolV"], t["g_poolX"], t.get("node_attr"))
            for k in ("g_h", "g_x", "g_A", "g_aggm", "g_aggx", "g_svel", "g_Bc"):
                t[k].copy_(r[k])
            if r["g_sgrav"] is not None:
                t["g_sgrav"].copy_(r["g_sgrav"])
            t["g_Zp"].copy_(r["g_Z"])
        elif name == "graph_pre_backward":
            xbar, mX, _ = self._graph_pre(w, t, cnt)
            mz = t["Z"] - xbar.unsqueeze(-1)
            Cn = t["Z"].size(2)
            g2 = t["g_Bc"].reshape(-1, H)
            dV1 = G["L.edge_mlp_virtual.0.weight"]
            dV1[:, H:2 * H] += g2.T @ t["HvT"].reshape(-1, H)
            dV1[:, 2 * H + 1:] += g2.T @ mX.transpose(1, 2).reshape(-1, Cn)
            G["L.edge_mlp_virtual.0.bias"] += g2.sum(0)
            t["g_HvT"] += t["g_Bc"] @ w.V1b
            g_mX = (t["g_Bc"] @ w.V1d).transpose(1, 2)
            g_mz = torch.einsum("bkd,bcd->bkc", mz, g_mX + g_mX.transpose(1, 2))
            t["g_Z"] += g_mz + t["g_Zp"]
            t["g_xbar"][:, :3] = -g_mz.sum(-1) / cnt.unsqueeze(1)
        elif name == "edge_backward":
            src = t["QX_src"]
            g_P, g_Q, (g_xr, g_xs) = F.edge_bwd(w, cfg, G, "L", graph.csr, t["P"], src[:, :H], t["x"], t["ea_sorted"],
                                                t["g_aggm"], t["g_aggx"], x_src=src[:, H:H + 3])
            t["g_P"].copy_(g_P); t["g_xrow"].copy_(g_xr)
            if not flags & F_GQX_ACCUM:      # the first launch of a layer zeroes the col-keyed sums, a later one adds to them
                t["g_QX_src"].zero_()
            t["g_QX_src"][:, :H] += g_Q
            t["g_QX_src"][:, H:H + 3] += g_xs
        elif name == "edge_col_reduce":
            pass                             # (the atomic form of the product path: edge_backward has already summed)
        elif name == "node_pre_backward":
            gq = t["g_QX"]
            g_h = F.node_pre_bwd(w, cfg, G, "L", t["h"], t["g_P"], gq[:N, :H], t["g_A"], t["g_svel"],
                                 t["g_sgrav"] if grav is not None else None)
            t["g_h"] += g_h
            t["g_x"] += t["g_xrow"] + gq[:N, H:H + 3] + t["g_xbar"][batch, :3]
            t["g_vel"] += t["svel"].unsqueeze(1) * t["g_x_out"]
        else:
            raise KeyError(name)

    def _graph_pre(self, w, t, cnt):
        xbar = t["xsum"][:, :3] / cnt.unsqueeze(1)
        mz = t["Z"] - xbar.unsqueeze(-1)
        mX = torch.einsum("bkc,bkd->bcd", mz, mz)
        Bc = t["HvT"] @ w.V1b.T + mX.transpose(1, 2) @ w.V1d.T + w.c1
        return xbar, mX, Bc
