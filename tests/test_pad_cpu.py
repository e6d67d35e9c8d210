"""CPU: the zero-padding that lets a hidden_nf < 64 model run on the 64-wide kernels (layout: fastegnn_amd/model.py:_pad_layout; the product pads with the HIP kernel behind fastegnn_pad_params)
preserves the reference function -- the oracle evaluated with the true hidden_nf equals the oracle evaluated at
hidden_nf = 64 on the padded parameters (models/FastEGNN.py:28-99 block layouts), and gradients slice back."""
import pytest
import torch

from tests.helpers import pad_reference as _pad_param
from oracle import fastegnn_ref as R


@pytest.mark.parametrize("h,C,na", [(20, 3, 2), (32, 1, 0), (1, 2, 1)])
def test_padded_parameters_compute_the_same_function(h, C, na):
    kw = dict(n_layers=2, gravity=[0, -1, 0], attention=True)
    cfg, cfg64 = R.Config(2, na, 2, h, C, **kw), R.Config(2, na, 2, 64, C, **kw)
    p = {k: v.double().requires_grad_(True) for k, v in R.init_params(cfg, seed=3).items()}
    pp = {k: _pad_param(k, v, h, C, False) for k, v in p.items()}
    for k, v in R.init_params(cfg64, seed=3).items():
        assert pp[k].shape == v.shape, k
    g = torch.Generator().manual_seed(0)
    N, E = 40, 200
    inp = dict(node_feat=torch.rand(N, 2, generator=g).double(), node_loc=torch.randn(N, 3, generator=g).double(),
               node_vel=torch.randn(N, 3, generator=g).double(), edge_index=torch.randint(0, N, (2, E), generator=g),
               data_batch=torch.cat([torch.zeros(25), torch.ones(15)]).long(),
               loc_mean=torch.randn(2, 3, C, generator=g).double(), edge_attr=torch.rand(E, 2, generator=g).double(),
               node_attr=torch.rand(N, na, generator=g).double() if na else None)
    a = R.forward({k: v.detach() for k, v in p.items()}, cfg, **inp)
    b = R.forward(pp, cfg64, **inp)
    assert (a[0] - b[0]).abs().max() < 1e-12 and (a[1] - b[1]).abs().max() < 1e-12
    (b[0].pow(2).sum() + b[1].pow(2).sum()).backward()          # gradients arrive in the reference's shapes
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    a = R.forward(q, cfg, **inp)
    (a[0].pow(2).sum() + a[1].pow(2).sum()).backward()
    for k in p:
        if q[k].grad is None:
            assert p[k].grad is None or p[k].grad.abs().max() == 0, k
        else:
            assert (p[k].grad - q[k].grad).abs().max() <= 1e-9 * (1 + q[k].grad.abs().max()), k


def test_padded_fastrf_parameters_compute_the_same_function():
    """Same algebra for the FastRF sibling (models/FastRF.py): its coord_mlp_vel.0 takes the 1-wide velocity norm, which
    is not a hidden-sized block (rf=True in _pad_param)."""
    import fastegnn_amd
    from oracle import fastrf_ref as RF
    h, C = 24, 2
    torch.manual_seed(5)
    m = fastegnn_amd.FastRF(2, 0, 2, h, C, n_layers=2, attention=True, gravity=[0, -1, 0])
    p = {k: v.detach().double() for k, v in m.state_dict().items()}
    pp = {k: _pad_param(k, v, h, C, True) for k, v in p.items()}
    m64 = fastegnn_amd.FastRF(2, 0, 2, 64, C, n_layers=2, attention=True, gravity=[0, -1, 0])
    for k, v in m64.state_dict().items():
        assert pp[k].shape == v.shape, k
    kw = dict(n_layers=2, gravity=[0, -1, 0], attention=True)
    cfg, cfg64 = R.Config(2, 0, 2, h, C, **kw), R.Config(2, 0, 2, 64, C, **kw)
    g = torch.Generator().manual_seed(1)
    N, E = 30, 120
    inp = dict(node_feat=torch.rand(N, 2, generator=g).double(), node_loc=torch.randn(N, 3, generator=g).double(),
               node_vel=torch.randn(N, 3, generator=g).double(), edge_index=torch.randint(0, N, (2, E), generator=g),
               data_batch=torch.cat([torch.zeros(18), torch.ones(12)]).long(),
               loc_mean=torch.randn(2, 3, C, generator=g).double(), edge_attr=torch.rand(E, 2, generator=g).double())
    a = RF.forward(p, cfg, **inp)
    b = RF.forward(pp, cfg64, **inp)
    assert (a[0] - b[0]).abs().max() < 1e-12 and (a[1] - b[1]).abs().max() < 1e-12
