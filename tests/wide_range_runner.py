"""Child process of tests/test_gpu_wide_range.py: one golden through the HIP module with its hidden features scaled up, under
the library the environment selects (FASTEGNN_WIDE_RANGE, FASTEGNN_DEBUG_CHECKS); prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    scale = float(sys.argv[1])
    from fastegnn_amd import _lib as K
    from oracle import fastegnn_ref as R
    from tests.gpu_util import model_from_golden
    from tests.helpers import Golden, rel_err
    g = Golden("c16_two_graphs")
    m = model_from_golden(g, device="cuda")
    with torch.no_grad():      # hidden features of ~scale: beyond fp16's range for scale >> 65 504
        m.embedding_in.weight.mul_(scale)
        m.embedding_in.bias.mul_(scale)
    kw, target, wv = g.model_kwargs(device="cuda")
    out = dict(lib=os.path.basename(K.LIB_PATH), raised=None)
    try:
        loc, vloc = m(**kw)
        out["finite"] = bool(torch.isfinite(loc).all() and torch.isfinite(vloc).all())
        p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        rl, rv = R.forward(p, g.cfg, **{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in kw.items()})
        out["ref_finite"] = bool(torch.isfinite(rl).all())
        out["err_loc"] = rel_err(loc.detach().cpu(), rl) if out["finite"] else None
    except FloatingPointError as e:
        out["raised"] = str(e)[:80]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
