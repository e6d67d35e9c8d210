"""Child process of tests/test_gpu_wide_range.py: one golden through the HIP module with its hidden features scaled up, under
the library policy the environment selects (FASTEGNN_WIDE_RANGE unset / 0 / 1); prints one JSON line.

    python -m tests.wide_range_runner <scale> [golden name]
"""
import json
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    scale = float(sys.argv[1])
    name = sys.argv[2] if len(sys.argv) > 2 else "c16_two_graphs"
    from fastegnn_amd import _lib as K
    from oracle import fastegnn_ref as R
    from tests.gpu_util import model_from_golden
    from tests.helpers import Golden, golden_loss, rel_err
    g = Golden(name)
    m = model_from_golden(g, device="cuda")
    with torch.no_grad():      # hidden features of ~scale: beyond fp16's range for scale >> 65 504
        m.embedding_in.weight.mul_(scale)
        m.embedding_in.bias.mul_(scale)
    kw, target, wv = g.model_kwargs(device="cuda")
    out = dict(lib=os.path.basename(K.LIB_PATH), raised=None, warned=False)
    try:
        sync_mode = os.environ.get("FASTEGNN_RANGE_CHECK", "deferred") == "sync"
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            loc, vloc = m(**kw)
            golden_loss(loc, vloc, target, wv).backward()
            out["first_finite"] = bool(torch.isfinite(loc).all() and torch.isfinite(vloc).all())
            # the parameter gradients of a pass that overflowed are ZEROED on the device (never NaN), whenever the host hears of it
            g1 = [q.grad for q in m.parameters() if q.grad is not None]
            out["first_grads_finite"] = all(bool(torch.isfinite(t).all()) for t in g1)
            out["first_grads_zero"] = all(float(t.abs().max()) == 0.0 for t in g1)
            if not out["first_finite"] and not sync_mode:
                # deferred check (the default): nothing waited for the guard launch; the NEXT call polls the host-mapped word,
                # switches the build and is what a caller of the reference would have got
                torch.cuda.synchronize()
                m.zero_grad(set_to_none=True)
                loc, vloc = m(**kw)
                golden_loss(loc, vloc, target, wv).backward()
            loc2, _ = m(**kw)                  # a further call stays on the build the last one ended on, silently
        out["warned"] = sum("wide-range" in str(w.message) for w in wlist)
        out["wide"] = bool(m._range.wide)
        out["finite"] = bool(torch.isfinite(loc).all() and torch.isfinite(vloc).all() and torch.isfinite(loc2).all())
        p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        rl, rv = R.forward(p, g.cfg, **{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in kw.items()})
        out["ref_finite"] = bool(torch.isfinite(rl).all())
        out["err_loc"] = rel_err(loc.detach().cpu(), rl.detach()) if out["finite"] else None
        if out["finite"] and out["ref_finite"]:
            golden_loss(rl, rv, target.cpu(), wv.cpu() if torch.is_tensor(wv) else wv).backward()
            gm = dict(m.named_parameters())
            errs = {}
            for k in ("embedding_in.weight", "gcl_0.edge_mlp.0.weight", "gcl_1.edge_mlp_virtual.2.weight", "virtual_node_feat"):
                if gm[k].grad is not None and p[k].grad is not None:
                    errs[k] = rel_err(gm[k].grad.cpu(), p[k].grad)
            out["grad_finite"] = all(bool(torch.isfinite(q.grad).all()) for q in gm.values() if q.grad is not None)
            out["err_grad_max"] = max(errs.values()) if errs else None
    except FloatingPointError as e:
        out["raised"] = str(e)[:160]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
