"""CPU: the oracle's restatement of the harness loss (MSE + MMD, utils/train.py:104-165) and of Adam
(main_nbody.py:137) against the training-step goldens captured from the reference
(oracle/gen_goldens.py --train-only: reference model + utils.train.kernel + torch.optim.Adam)."""
import numpy as np
import pytest
import torch

from oracle import fastegnn_ref as R
from tests.helpers import GOLDEN_DIR, rel_err

NAMES = ["train_nbody5", "train_ragged_simulation"]


def load(name):
    z = np.load(f"{GOLDEN_DIR}/{name}.npz")
    g = {}
    for k in z.files:
        a, b = k.split("/", 1)
        g.setdefault(a, {})[b] = torch.from_numpy(np.asarray(z[k]))
    m = g["meta"]
    cfg = R.Config(2, 0, 2, 64, int(m["C"]), n_layers=int(m["L"]))
    return g, cfg


@pytest.mark.parametrize("name", NAMES)
def test_oracle_training_steps_match_reference(name):
    g, cfg = load(name)
    m = g["meta"]
    inp = {k: v for k, v in g["in"].items() if k not in ("loc_t", "sample_nodes")}
    p = {k: v.clone() for k, v in g["p0"].items()}
    state = {}
    for step in range(1, 4):
        pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        loc, vloc = R.forward(pp, cfg, **inp)
        loss, mse = R.loss_mse_mmd_nodes(loc, vloc, g["in"]["loc_t"], g["in"]["sample_nodes"], float(m["sigma"]),
                                         float(m["weight"]))
        loss.backward()
        assert abs(loss.item() - float(g["out"]["losses"][step - 1, 0])) < 2e-6
        assert abs(mse.item() - float(g["out"]["losses"][step - 1, 1])) < 2e-6
        grads = {k: v.grad for k, v in pp.items()}
        if step == 1:
            assert rel_err(loc, g["out"]["loc"]) < 2e-6 and rel_err(vloc, g["out"]["vloc"]) < 2e-6
            for k, v in g["g1"].items():
                got = grads[k] if grads[k] is not None else torch.zeros_like(v)
                assert rel_err(got, v) < 5e-5, k
        R.adam_step(p, grads, state, step, lr=float(m["lr"]), weight_decay=float(m["wd"]))
        if step in (1, 3):
            ref = g[f"p{step}"]
            bad = sum(int(((p[k] - ref[k]).abs() > 2e-5).sum()) for k in p)
            tot = sum(v.numel() for v in p.values())
            assert bad <= 1e-4 * tot, (step, bad, tot)     # sign flips of gradients that are pure rounding noise
