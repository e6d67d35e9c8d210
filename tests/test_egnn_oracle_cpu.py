"""CPU: the EGNN-baseline oracle (oracle/egnn_ref.py) against goldens captured from the reference's
models/basic.py EGNN (oracle/gen_goldens.py --egnn), outputs and all gradients."""
import numpy as np
import pytest
import torch

from oracle import egnn_ref as E
from tests.helpers import GOLDEN_DIR, rel_err

EGNN_NAMES = ["egnn_with_v", "egnn_no_v", "egnn_clamped", "egnn_norm"]   # egnn_norm: EGNN(norm=True), basic.py:271-272
# the sibling's wide path (fastegnn_amd/wide.py): flat=True (every MLP Tanh with 4 x hidden inner units, basic.py:176-178) and
# hidden_nf = 128; oracle/gen_goldens.py --egnn-wide
EGNN_WIDE_NAMES = ["egnn_flat", "egnn_h128"]


def egnn_act(g):
    """the activation the golden's model evaluates: flat=True replaces it by Tanh (the widths are in the weights)"""
    return torch.tanh if int(g["meta"].get("flat", 0)) else torch.nn.functional.silu


def load_egnn(name):
    z = np.load(f"{GOLDEN_DIR}/{name}.npz")
    g = {}
    for k in z.files:
        a, b = k.split("/", 1)
        g.setdefault(a, {})[b] = torch.from_numpy(np.asarray(z[k]))
    return g


def egnn_loss(x, h, target, wh):
    return torch.nn.functional.mse_loss(x, target) + 0.05 * (h * wh).sum() / x.size(0)


@pytest.mark.parametrize("name", EGNN_NAMES + EGNN_WIDE_NAMES)
def test_egnn_oracle_matches_reference(name):
    g = load_egnn(name)
    p = {k: v.clone().requires_grad_(True) for k, v in g["p"].items()}
    i = g["in"]
    leaf = {k: i[k].clone().requires_grad_(True) for k in ("x", "h") + (("v",) if "v" in i else ())}
    x, h = E.forward(p, int(g["meta"]["L"]), leaf["x"], leaf["h"], i["edge_index"], i["edge_fea"], leaf.get("v"),
                     norm=bool(int(g["meta"].get("norm", 0))), act=egnn_act(g))
    assert rel_err(x, g["out"]["x"]) < 2e-6 and rel_err(h, g["out"]["h"]) < 2e-5
    egnn_loss(x, h, i["target"], i["wh"]).backward()
    for k, v in p.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert rel_err(got, g["gp"][k]) < 5e-5, k
    for k, v in leaf.items():
        assert rel_err(v.grad, g["gin"][k]) < 5e-5, k
