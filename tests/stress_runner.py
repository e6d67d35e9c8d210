"""Run-to-run repeat check of the HIP path (imported by tests/test_gpu_stress.py; also `python -m tests.stress_runner`).

The same forward+backward is run `iters` times on one seeded input; every output and every gradient of every pass is
compared with pass 0.  The kernels are deterministic up to the summation order of float atomics (the per-graph pools,
the centroid sums) and of the ticket-ordered in-workgroup weight gradients, so two passes differ by rounding noise only:
measured <= 3e-7 on the outputs and <= 2e-5 of max|g| on every gradient but the strongly cancelling layer-0
`coord_mlp_v_virtual` sums (<= 2e-3).  The limits below are 3-10x those levels; the failure this test exists for
(DESIGN.md, round-2 ledger: one pytest run with 1e-2 .. 1e-1 errors in edge_mlp.0.weight / virtual_node_feat) is far
outside them.  Under FASTEGNN_SAFE_WAITS=1 the same check runs on the conservative-synchronisation build.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

OUT_LIMIT = 2e-6          # loc / vloc, relative to max|.|
GRAD_LIMIT = 1e-4         # gradients, relative to max|g| of the tensor
NOISY = ("coord_mlp_v_virtual", "coord_mlp_r_virtual")   # cancelling sums over all (node, channel) rows
NOISY_LIMIT = 1e-2


def _rel(a, b):
    d = (a - b).abs().max().item()
    m = b.abs().max().item()
    return d / m if m > 0 else d


def _limit(name):
    if name in ("loc", "vloc"):
        return OUT_LIMIT
    return NOISY_LIMIT if any(s in name for s in NOISY) else GRAD_LIMIT


def cases():
    import fastegnn_amd
    from bench import make_frame
    from tests.test_gpu_properties import _batch
    torch.manual_seed(0)
    # the headline configuration at the size of test_cfg4_headline_shape_vs_oracle
    m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
    frame, target = make_frame(20000, 16, 7, "cuda")
    yield "headline20k", m, frame, target
    # the many-tiles shape (test_many_tiles_per_workgroup_vs_oracle): three graphs, C=8, nine tiles per workgroup
    m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 8, device="cuda", n_layers=2, gravity=[0, -1, 0])
    inp = {k: v.cuda() for k, v in _batch([20000, 12000, 4900], 2, 8, seed=13).items()}
    yield "manytiles", m, inp, inp["node_loc"] + 0.3
    # many tiny graphs with attention: every tile crosses graph boundaries (per-row atomic paths)
    m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 3, device="cuda", n_layers=4, attention=True)
    inp = {k: v.cuda() for k, v in _batch([5] * 100, 3, 3, seed=3).items()}
    yield "tiny100x5", m, inp, inp["node_loc"] + 0.3


def run_case(name, model, frame, target, iters, dump=None):
    ref, worst, bad = None, {}, []
    for it in range(iters):
        for p in model.parameters():
            p.grad = None
        loc, vloc = model(**frame)
        (torch.nn.functional.mse_loss(loc, target) + 0.1 * vloc.square().mean()).backward()
        cur = {"loc": loc.detach().clone(), "vloc": vloc.detach().clone()}
        cur.update({k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None})
        if not all(torch.isfinite(v).all() for v in cur.values()):
            bad.append((it, "non-finite", float("nan")))
            continue
        if ref is None:
            ref = cur
            if dump:
                np.savez(dump, **{k: v.cpu().numpy() for k, v in ref.items()})
            continue
        for k in ref:
            e = _rel(cur[k], ref[k])
            worst[k] = max(worst.get(k, 0.0), e)
            if e > _limit(k):
                bad.append((it, k, e))
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    return dict(case=name, iters=iters, bad=bad[:20], n_bad=len(bad), worst={k: v for k, v in top})


def main(argv):
    iters = int(argv[1]) if len(argv) > 1 else 100
    dump_dir = argv[2] if len(argv) > 2 else None
    from fastegnn_amd import _lib
    res = dict(safe_waits=_lib.SAFE_WAITS, lib=os.path.basename(_lib.LIB_PATH), cases=[])
    for name, m, frame, target in cases():
        dump = os.path.join(dump_dir, f"{name}.npz") if dump_dir else None
        res["cases"].append(run_case(name, m, frame, target, iters, dump))
    print("STRESS " + json.dumps(res))
    return 1 if any(c["n_bad"] for c in res["cases"]) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
