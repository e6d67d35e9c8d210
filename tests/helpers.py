"""Shared helpers for the parity tests (golden loading, error metrics)."""
import glob
import os

import numpy as np
import torch

from oracle import fastegnn_ref as R

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(include_fp64=False, silu_only=False):
    """silu_only: without the act_* goldens (act_fn other than SiLU) -- for the stage-level mirror oracle/factored.py,
    which restates the kernels' factorisation for the default activation only"""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    if silu_only:
        names = [n for n in names if not n.startswith("act_")]
    names = [n for n in names if not n.startswith(("train_", "egnn_", "dataset_", "fastrf_", "wide_"))]   # other fixture families have their own tests
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
            gravity=[float(v) for v in m["gravity"]] if int(m["has_gravity"]) else None,
            act=str(m["act"]) if "act" in m else "silu", act_param=float(m["act_param"]) if "act_param" in m else 0.0)

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

# Gradients and the displacement are compared through the fp64 oracle (hard part 2): the
# build's error against exact arithmetic may be GRAD_FACTOR x the fp32 reference's own error on the same tensor,
# plus GRAD_FLOOR (both as max-error relative to max|truth| of the tensor).
GRAD_FACTOR = 2.0
GRAD_FLOOR = 1e-6
# Explicit per-tensor exceptions: (case pattern, tensor pattern) -> (factor, floor).  Generated from a full -m gpu
# run with FASTEGNN_TOL_DUMP=<file> (every comparison is logged; tools/tol_report.py prints the offenders) --
# each entry states what was measured.  A tensor lands here when its gradient is a short or strongly cancelling
# sum, for which two fp32 evaluations in different summation orders (the reference's ATen kernels, these HIP
# kernels) each sit at a random point of the same rounding-noise band, so "2 x the reference's own draw" is not a
# bound; the floor granted is the measured band, never more than 1.5e-5.
GRAD_EXCEPTIONS = [
    # Round 5: the entry for the input gradients of the 17-node activation goldens ("^act_", "^gin/", floor 2.5e-6) is gone: graphs of
    # at most 512 edges take the order-independent col-keyed sum by default (FastEGNN.deterministic_for), and over five runs on the
    # default and on the wide-range build no such comparison exceeds the plain rule (worst 1.59e-6 / 1.37e-6: gpurun_out/actgin, tools/gpu_r5_act_gin.sh)
    # Round 4 (f16x2 products, double-precision slab reduce / bias totals / embedding sums; profiles/r04_gradient_tolerance_report.txt:
    # 4 385 comparisons over the whole -m gpu suite, 27 beyond 2 x ref + 1e-6, none beyond what is granted here): the entries for
    # the layer-0 bias sums of the virtual coordinate heads on the cfg4 / cfg5 shapes (factor 4 until round 3: now 0.35-0.69 of
    # the PLAIN tolerance over four runs on four boxes) and for the no-edges case are gone; floors of the edge stage, the
    # embedding bias, the scalar head biases and the virtual coordinate head tightened to ~2 x what was measured.
    # (case regex, tensor regex, factor, floor, why) -- re-measured at the end of round 3 on an MI355X with FASTEGNN_TOL_DUMP
    # over the whole -m gpu suite (4 054 comparisons over 45 cases, 29 beyond 2 x ref + 1e-6; the operand split of the
    # bf16x3 products rounds to nearest since then, which took the attention goldens' excesses down 2-4 x): every entry
    # states what was measured and grants ~1.3-1.5 x that
    (r"ragged3_(allflags|normalize)|fastrf_allflags", r".", 3.0, 1e-6,
     "normalize=True on graphs with self loops / coincident points: d/(|d|+1e-8) at d = 0 amplifies rounding noise by "
     "1e8, the reference's own gradients are 5e-4..5e-3 from exact arithmetic there, and a mathematically identical "
     "fp32 re-association run on the SAME torch CPU kernels (oracle/factored.py in tests/test_factored_cpu.py) "
     "measures 2.1-5.7x the reference's draw on 13 tensors; HIP: <= 2.36x (gin/node_vel 1.13e-2 vs 4.81e-3; factor 8 "
     "until round 3)"),
    (r"cpu_reassociation:.*(ragged3_(allflags|normalize|attention)|fastrf_allflags)", r".", 8.0, 1e-6,
     "tests/test_factored_cpu.py only: the mathematically identical fp32 re-association on the torch CPU kernels "
     "(oracle/factored.py) itself measures 2.1-5.7x the reference's draw on 13 tensors of the normalize goldens and up "
     "to 10.8x on att_mlp.0.bias -- the yardstick that says these goldens are ill-conditioned, not a product tolerance"),
    (r"cpu_reassociation:.*(attention|allflags)", r"att_mlp(_virtual)?\.0\.(weight|bias)", 25.0, 1e-6,
     "same, attention gates (10.8x measured)"),
    (r"ragged3_attention|ragged3_allflags", r".", 4.0, 1e-6,
     "the attention goldens are moderately ill-conditioned (reference 1e-5 from exact arithmetic on the layer-1 edge "
     "stage, 10x its usual level): HIP measures <= 3.3x the reference's draw on the edge-stage tensors of that layer "
     "(gcl_1.coord_mlp_r.0.bias 3.59e-6 vs 1.09e-6; 2.5-6.0x and factor 8 with the truncating split of rounds 1-2)"),
    (r"attention|allflags", r"att_mlp(_virtual)?\.0\.(weight|bias)", 12.0, 8e-6,
     "attention gates: scalar / 64-vector gradients that are cancelling sums over ~100 edges whose per-edge term "
     "g_a = <g_m, m0> is itself a cancelling 64-term dot product; the CPU re-association above measures 10.8x the "
     "reference's draw on att_mlp.0.bias (9.1e-6 vs 8.4e-7); HIP: 8.0x (6.77e-6, ragged3_attention gcl_1), 5.65e-6 "
     "against a reference draw of 1.4e-8 on the 4 000-node GELU case (23.5-33x and factor 40 with the truncating split))"),
    (r".", r"(edge_mlp|coord_mlp_r|edge_message_net\.scalar_net\.mlp|coord_net\.mlp)\.", 2.0, 1e-5,
     "parameter gradients of the edge stage: sums over up to 370 k edges with cancellation (max|g| ~1e-8 on the last "
     "layers of the radius-graph cases): measured <= 1.02e-5 over 2 x ref (cfg5 shape at 20 k nodes, "
     "gcl_3.edge_mlp.0.bias 1.39e-5 vs ref 1.87e-6) where the reference sits at 1e-6..6e-6.  NOT the transcendentals: one "
     "Newton step on the sigmoid's reciprocal and an exp2 argument corrected to < 1 ulp, in the backward recompute, were "
     "both measured (profiles/r03_lever_*.txt) and move this figure by < 5 %; the bf16x3 split's rounding mode moves it "
     "by 10 % -- what remains is the order of the fp32 sums (floor 2e-5 until round 3, 1.5e-5 in round 3; round 4 with f16x2 "
     "products and double-precision slab sums: <= 5.1e-6 over 2 x ref in four runs, gcl_3.edge_mlp.0.bias 8.83e-6 vs 1.87e-6)"),
    (r"wide_", r".", 2.0, 2e-6,
     "round 6, the hidden_nf > 64 path (tests/test_gpu_wide.py): every edge- and node-sized sum of its backward ends in fp32 atomics in arrival "
     "order, so -- unlike the fused path, whose comparisons repeat bit for bit -- each comparison moves by +- 0.05 .. 0.15 of its tolerance between "
     "runs of one binary (three passes of the suite on one box, tools/gpu_r6_margin.sh + tools/tol_margin.py: wide_h256 "
     "gcl_0.node_mlp_virtual.0.bias 0.81 / 0.92 / 0.87 of the plain tolerance, wide_h256 gcl_0.node_mlp_virtual.2.bias 0.59 / 0.77 / 0.62, wide_ceilings "
     "gcl_0.coord_mlp_v_virtual.0.bias 0.21 / 0.62 / 0.48; wide_h160 gcl_0.att_mlp.0.bias 0.88 / 0.93 in 12 repeats and beyond 1.0 once in a "
     "full-suite run, tools/gpu_r6_wide_repeat.sh).  Floor 2e-6 in place of 1e-6 for the path puts the worst of them at 0.71: the means are "
     "inside the plain rule, the floor pays for the run-to-run band"),
    (r"at_52k_nodes", r"(edge_mlp|coord_mlp_r)\.", 2.0, 5e-5,
     "round 6, the first oracle comparison at ~1 M edges (52 000 nodes of the cfg4 shape: tests/test_gpu_virt_cs.py): the parameter gradients "
     "of the edge stage are cancelling sums over every edge (max|g| ~1e-9) whose error grows with the edge count -- per-edge gradients with an error of "
     "2^-23 of the item's largest component (f16x2 item scaling) and 768-row fp32 register chains: profiles/r06_edge_grad_accuracy.txt -- where the "
     "floor above was calibrated at <= 370 k edges: gcl_3.edge_mlp.0.bias 2.99e-5, "
     "gcl_3.edge_mlp.0.weight 3.03e-5, gcl_0.coord_mlp_r.0.weight 2.50e-5 against a reference at 2.0e-6 .. 4.6e-6 (one run; every other "
     "tensor of the case, the whole virtual stage included, passes the plain rule).  The atomics of the edge backward arrive in another order every run: "
     "gcl_3.edge_mlp.0.bias 3.39e-5 / 3.01e-5 / 2.86e-5 over three runs on one box (tools/gpu_r6_margin.sh), i.e. AT a floor of 3e-5 -- the floor is 5e-5, "
     "1.5 x the band's top.  A finding, not a target: DESIGN.md section 6"),
    (r".", r"embedding_in\.bias", 2.0, 6e-6,
     "the column sum of the gradient that leaves the first layer -- every rounding of the whole backward chain ends in "
     "it: 3.4e-6 / 2.7e-6 against the reference's 1.1e-6 / 7.9e-7 (nbody5_cfg1_trained, train_ragged_simulation); round 4, with "
     "double accumulators in the embedding's weight-gradient kernel: 1.51e-6 over 2 x ref (nbody5_cfg1: 6.36e-6 vs 2.42e-6) -- the "
     "error sits in g_h, not in its column sum.  Round 6: the software-pipelined f16x2 products add their low-part sum LAST ((acc + hh) "
     "+ lo / 2^11 instead of (acc + lo / 2^11) + hh: the same terms, one rounding in another place): nbody5_cfg1 7.88e-6 with, 7.54e-6 "
     "without, bit-identical over six runs each (gpurun_out/r6h) = 3.04e-6 over 2 x ref; with two waves per tile in virt_fwd (the node-MLP "
     "accumulator of a tile is then the sum of two partial sums) 9.77e-6 = 4.93e-6 over 2 x ref.  Every rounding-level change of the forward moves "
     "this one cancelling scalar-per-feature sum by more than the reference's own draw: floor 3e-6 -> 6e-6"),
    (r".", r"(gravity_mlp|coord_mlp_vel)\.2\.bias", 2.0, 2.5e-6,
     "scalar head biases: one number, the sum of N per-node terms: 2.44e-6 against the reference's 3.7e-7 "
     "(nbody5_cfg1_trained, gcl_1.coord_mlp_vel.2.bias), 3.28e-6 against 1.10e-6 (act_softplus, gcl_0.gravity_mlp.2.bias)"),
    (r".", r"coord_mlp_v_virtual\.|att_mlp_virtual\.", 2.0, 5e-6,
     "the virtual coordinate head and gate: [1,64] / scalar gradients summed per (tile, channel) over the tile first (DPP) and "
     "then across tiles in LDS -- a different association of a cancelling sum: 5.81e-6 over 2 x ref "
     "(ragged3_attention, gcl_0.att_mlp_virtual.0.bias 9.60e-6 vs 1.89e-6; 1.26e-5 / 1.36e-5 and floor 1e-5 with the "
     "truncating split)"),
    (r"act_mid", r"\.bias$", 2.0, 5e-6,
     "activations other than SiLU at 4 000 nodes: bias gradients are column sums over 32 k - 48 k rows behind erf / exp / "
     "log1p evaluations of 2-4 ulp; measured 4.64e-6 against 2 x ref + 1e-6 = 2.96e-6 (act_mid_gelu, "
     "gcl_1.edge_mlp_virtual.0.bias)"),
]


def grad_tolerance(case, name, e_ref):
    import re
    f, fl = GRAD_FACTOR, GRAD_FLOOR
    for cre, nre, ef, efl, _why in GRAD_EXCEPTIONS:
        if re.search(cre, case) and re.search(nre, name):
            f, fl = max(f, ef), max(fl, efl)
    return f * e_ref + fl


_DUMP = os.environ.get("FASTEGNN_TOL_DUMP")


def grad_check(case, name, got, ref32, truth, bad):
    """Appends a message to `bad` when `got` is further from `truth` (fp64) than the calibrated tolerance allows.
    With FASTEGNN_TOL_DUMP=<file> every comparison is logged as a JSON line and nothing fails (calibration run)."""
    e_ref, e_got = rel_err(ref32, truth), rel_err(got, truth)
    tol = grad_tolerance(case, name, e_ref)
    if _DUMP:
        import json
        with open(_DUMP, "a") as f:
            f.write(json.dumps({"case": case, "tensor": name, "got": e_got, "ref": e_ref, "tol": tol,
                                "max": float(torch.as_tensor(truth).abs().max()), "numel": int(torch.as_tensor(truth).numel())}) + "\n")
        return
    if e_got > tol:
        bad.append(f"{name} {e_got:.2e} (ref {e_ref:.2e}, tol {tol:.2e})")


def check_parity(g, loc, vloc, G=None, gin=None, truth=None, case_prefix=""):
    """loc/vloc: <=1e-5 rel of the reference golden.  Displacement and gradients: the fp32
    reference itself is 1e-7..1e-3 away from exact arithmetic depending on the tensor, so the
    build's error against the fp64 oracle must stay within GRAD_FACTOR x the reference's own
    fp32 error (+ GRAD_FLOOR), see grad_check."""
    t_loc, t_vloc, t_G, t_gin = truth if truth is not None else fp64_truth(g)
    msgs = []
    e = rel_err(loc, g.out["loc"]); msgs += [f"loc {e:.2e}"] if e >= OUT_TOL else []
    e = rel_err(vloc, g.out["vloc"]); msgs += [f"vloc {e:.2e}"] if e >= OUT_TOL else []
    x0 = torch.from_numpy(g.inp["node_loc"]).double()
    grad_check(case_prefix + g.name, "displacement", torch.as_tensor(loc).double().cpu() - x0,
               torch.from_numpy(g.out["loc"]).double() - x0, t_loc - x0, msgs)
    for got, ref, tru, tag in ((G, g.gp, t_G, "gp"), (gin, g.gin, t_gin, "gin")):
        if got is None:
            continue
        for k in ref:
            if k not in got or got[k] is None:
                msgs.append(f"{tag}/{k} missing")
                continue
            grad_check(case_prefix + g.name, f"{tag}/{k}", got[k], ref[k], tru[k], msgs)
    return msgs


def pad_reference(name, p, h, C_, rf):
    """Differentiable torch statement of the zero-padded 64-wide image of a narrow model's parameter, built from the
    layout the product uses (fastegnn_amd.model._pad_layout; the product's own implementation is the HIP kernel behind
    fastegnn_pad_params).  Checker for tests/test_pad_cpu.py and tests/test_gpu_properties.py."""
    import torch
    from fastegnn_amd.model import _pad_layout
    rows, cols, rows_dst, blocks, out_shape = _pad_layout(name, p.shape, h, C_, rf)
    w = p.reshape(rows, cols)
    pieces, at = [], 0
    for b in blocks:
        pieces.append(torch.nn.functional.pad(w[:, at:at + b], (0, b // h * (64 - h))))
        at += b
    pieces.append(w[:, at:])
    w = torch.cat(pieces, dim=1) if len(pieces) > 1 else pieces[0]
    w = torch.nn.functional.pad(w, (0, 0, 0, rows_dst - rows))
    return w.reshape(out_shape)
