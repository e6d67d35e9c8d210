"""Shared helpers for the parity tests (golden loading, error metrics)."""
import glob
import os

import numpy as np
import torch

from oracle import fastegnn_ref as R

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(include_fp64=False):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    if not include_fp64:
        names = [n for n in names if not n.endswith("_fp64")]
    return names


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.inp, self.params, self.out, self.gp, self.gin, self.meta = {}, {}, {}, {}, {}, {}
        for k in z.files:
            grp, rest = k.split("/", 1)
            t = z[k]
            {"in": self.inp, "p": self.params, "out": self.out, "gp": self.gp,
             "gin": self.gin, "meta": self.meta}[grp][rest] = t
        m = self.meta
        self.cfg = R.Config(
            node_feat_nf=int(m["nf"]), node_attr_nf=int(m["na"]), edge_attr_nf=int(m["ea"]),
            hidden_nf=int(m["H"]), virtual_channels=int(m["C"]), n_layers=int(m["L"]),
            residual=bool(m["residual"]), attention=bool(m["attention"]),
            normalize=bool(m["normalize"]), tanh=bool(m["tanh"]),
            gravity=[float(v) for v in m["gravity"]] if int(m["has_gravity"]) else None)

    def tensors(self, d, device="cpu", dtype=None):
        out = {}
        for k, v in d.items():
            t = torch.from_numpy(np.asarray(v))
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            out[k] = t.to(device)
        return out

    def model_kwargs(self, device="cpu", dtype=None):
        t = self.tensors(self.inp, device, dtype)
        kw = dict(node_feat=t["node_feat"], node_loc=t["node_loc"], node_vel=t["node_vel"],
                  edge_index=t["edge_index"], data_batch=t["data_batch"], loc_mean=t["loc_mean"],
                  edge_attr=t["edge_attr"], node_attr=t.get("node_attr"))
        return kw, t["target"], t["wv"]


def golden_loss(loc, vloc, target, wv):
    """The loss oracle/gen_goldens.py differentiates."""
    return torch.nn.functional.mse_loss(loc, target) + 0.3 * (vloc * wv).sum() / vloc.numel()


def rel_err(a, b):
    """max |a-b| / max(|b|) -- scale-relative max error (b is the reference)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)
