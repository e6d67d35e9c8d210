"""Shared helpers for the parity tests (golden loading, error metrics)."""
import glob
import os

import numpy as np
import torch

from oracle import fastegnn_ref as R

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(include_fp64=False):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    names = [n for n in names if not n.startswith(("train_", "egnn_", "dataset_", "fastrf_"))]   # other fixture families have their own tests
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


# --------------------------------------------------------------------------
# calibrated parity check (hard part 2)
# --------------------------------------------------------------------------
def fp64_truth(g):
    """Oracle in fp64 on the golden's fp32 weights/inputs: outputs and all gradients."""
    dt = torch.float64
    p = {k: v.clone().requires_grad_(True) for k, v in g.tensors(g.params, dtype=dt).items()}
    kw, target, wv = g.model_kwargs(dtype=dt)
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw.update(leaf)
    loc, vloc = R.forward(p, g.cfg, **kw)
    golden_loss(loc, vloc, target, wv).backward()
    G = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    gin = {k: v.grad for k, v in leaf.items()}
    return loc.detach(), vloc.detach(), G, gin


OUT_TOL = 1e-5        # north_star: outputs within 1e-5 rel fp32 of the reference
GRAD_FACTOR = 10.0    # build's fp32 error vs fp64 may be this multiple of the reference's own
GRAD_FLOOR = 2.5e-5   # fp32 noise of short cancelling sums: a 60-term scalar bias gradient (att_mlp.0.bias of
                      # ragged3_attention) sits at 1.0e-5 with fp32-input MFMA and 2.0e-5 with the bf16x3 products


def check_parity(g, loc, vloc, G=None, gin=None, truth=None):
    """loc/vloc: <=1e-5 rel of the reference golden.  Displacement and gradients: the fp32
    reference itself is 1e-7..1e-3 away from exact arithmetic depending on the tensor, so the
    build's error against the fp64 oracle must stay within GRAD_FACTOR x the reference's own
    fp32 error (+ floor)."""
    t_loc, t_vloc, t_G, t_gin = truth if truth is not None else fp64_truth(g)
    msgs = []
    e = rel_err(loc, g.out["loc"]); msgs += [f"loc {e:.2e}"] if e >= OUT_TOL else []
    e = rel_err(vloc, g.out["vloc"]); msgs += [f"vloc {e:.2e}"] if e >= OUT_TOL else []
    x0 = torch.from_numpy(g.inp["node_loc"]).double()
    d_t = t_loc - x0
    e_ref = rel_err(torch.from_numpy(g.out["loc"]).double() - x0, d_t)
    e_got = rel_err(torch.as_tensor(loc).double().cpu() - x0, d_t)
    if e_got > GRAD_FACTOR * e_ref + GRAD_FLOOR:
        msgs.append(f"displacement {e_got:.2e} (ref {e_ref:.2e})")
    for got, ref, tru, tag in ((G, g.gp, t_G, "gp"), (gin, g.gin, t_gin, "gin")):
        if got is None:
            continue
        for k in ref:
            if k not in got or got[k] is None:
                msgs.append(f"{tag}/{k} missing")
                continue
            e_ref = rel_err(ref[k], tru[k])
            e_got = rel_err(got[k], tru[k])
            if e_got > GRAD_FACTOR * e_ref + GRAD_FLOOR:
                msgs.append(f"{tag}/{k} {e_got:.2e} (ref {e_ref:.2e})")
    return msgs
