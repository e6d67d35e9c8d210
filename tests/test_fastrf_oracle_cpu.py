"""Row 8f-4 (FastRF): the CPU restatement (oracle/fastrf_ref.py) against goldens captured from the
reference's FastRF class -- outputs and every gradient through autograd."""
import pytest
import torch

from oracle import fastrf_ref as RF
from tests.helpers import Golden, golden_loss, rel_err

CASES = ["fastrf_plain", "fastrf_allflags", "fastrf_c16", "fastrf_h128"]   # fastrf_h128: hidden_nf = 128, the unfused wide path


@pytest.mark.parametrize("name", CASES)
def test_fastrf_oracle_matches_reference_golden(name):
    g = Golden(name)
    p = {k: v.clone().requires_grad_(True) for k, v in g.tensors(g.params).items()}
    kw, target, wv = g.model_kwargs()
    kw.pop("node_attr")
    leaf = {k: kw[k].clone().requires_grad_(True) for k in ("node_feat", "node_loc", "node_vel", "loc_mean")}
    kw.update(leaf)
    loc, vloc = RF.forward(p, g.cfg, **kw)
    assert rel_err(loc, g.out["loc"]) < 1e-6 and rel_err(vloc, g.out["vloc"]) < 1e-6
    golden_loss(loc, vloc, target, wv).backward()
    for k, ref in g.gp.items():
        got = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        assert rel_err(got, ref) < 1e-4, k   # two fp32 evaluations of small (1e-6) saturated-tanh gradients
    for k, ref in g.gin.items():
        got = leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k])
        assert rel_err(got, ref) < 1e-4, k   # two fp32 evaluations of small (1e-6) saturated-tanh gradients


def test_fastrf_state_dict_layout_matches_reference():
    import fastegnn_amd
    g = Golden("fastrf_allflags")
    m = fastegnn_amd.FastRF(2, 0, 2, 64, 4, n_layers=2, attention=True, normalize=True, tanh=True, gravity=[0, -1, 0])
    sd = m.state_dict()
    assert list(sd.keys()) == list(g.params.keys())
    assert all(tuple(sd[k].shape) == g.params[k].shape for k in sd)
    assert m.__class__.__name__ == "FastRF"     # the harness dispatches on it (utils/train.py:57)
    assert sd["gcl_0.coord_mlp_vel.0.weight"].shape == (64, 1) and not any("node_mlp" in k for k in sd)
