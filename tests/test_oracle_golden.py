"""CPU: pins the oracle (oracle/fastegnn_ref.py) against the golden vectors captured from the
real reference (oracle/gen_goldens.py; /root/reference/models/FastEGNN.py + autograd)."""
import pytest
import torch

from oracle import fastegnn_ref as R
from tests.helpers import Golden, golden_names, golden_loss, rel_err

# op-for-op restatement in the same dtype: differences are only op-fusion/ordering noise
TOL_OUT = 2e-6
TOL_GRAD = 2e-5


# (wide_h128_two_graphs: hidden_nf = 128, the golden of the unfused wide path -- tests/test_gpu_wide.py)
@pytest.mark.parametrize("name", golden_names(include_fp64=True) + ["wide_h128_two_graphs"])
def test_oracle_matches_reference_golden(name):
    g = Golden(name)
    dt = torch.float64 if name.endswith("_fp64") else torch.float32
    p = {k: v.clone().requires_grad_(True) for k, v in g.tensors(g.params, dtype=dt).items()}
    kw, target, wv = g.model_kwargs(dtype=dt)
    for k in ("node_feat", "node_loc", "node_vel", "loc_mean"):
        kw[k] = kw[k].clone().requires_grad_(True)
    loc, vloc, layers = R.forward(p, g.cfg, return_layers=True, **kw)
    assert rel_err(loc, g.out["loc"]) < TOL_OUT
    assert rel_err(vloc, g.out["vloc"]) < TOL_OUT
    # displacement is the informative quantity (coordinate heads are initialised tiny)
    disp = loc.detach() - kw["node_loc"].detach()
    disp_ref = torch.from_numpy(g.out["loc"]).to(dt) - kw["node_loc"].detach()
    assert rel_err(disp, disp_ref) < 2e-4
    for i, (h, x, Hv, Z) in enumerate(layers):
        assert rel_err(h, g.out[f"layer{i}/h"]) < 1e-5
        assert rel_err(x, g.out[f"layer{i}/x"]) < TOL_OUT
        assert rel_err(Hv, g.out[f"layer{i}/Hv"]) < 1e-5
        assert rel_err(Z, g.out[f"layer{i}/Z"]) < TOL_OUT
    loss = golden_loss(loc, vloc, target, wv)
    assert abs(loss.item() - float(g.out["loss"])) < 1e-5
    loss.backward()
    for k, v in p.items():
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        assert rel_err(gr, g.gp[k]) < TOL_GRAD, k
    for k in ("node_feat", "node_loc", "node_vel", "loc_mean"):
        assert rel_err(kw[k].grad, g.gin[k]) < TOL_GRAD, k


def test_init_params_has_reference_state_dict_layout():
    g = Golden("ragged3_allflags")
    p = R.init_params(g.cfg, seed=0)
    assert set(p.keys()) == set(g.params.keys())
    for k, v in p.items():
        assert tuple(v.shape) == tuple(g.params[k].shape), k
