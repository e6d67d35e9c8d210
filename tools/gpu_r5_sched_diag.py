"""diagnostic: FastEGNN(64) against the fp64 oracle on graphs whose edge count is / is not a multiple of the 16-edge tile, and whose
node count is / is not a multiple of 16 -- prints the worst gradient error of the edge-stage and of the other parameters."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fastegnn_ref as R
from tests.helpers import rel_err
from tests.test_gpu_properties import _batch, _loss, _models

for sizes, k in (([64], 4), ([65], 4), ([64, 32], 8), ([63, 33], 5), ([128], 16), ([130], 7), ([256], 16)):
    cfg = R.Config(2, 0, 2, 64, 3, n_layers=2)
    inp = _batch(sizes, k, 3, seed=3)
    p, m = _models(cfg, 3)
    tgt = inp["node_loc"] + 0.5
    kw = {kk: v.cuda() for kk, v in inp.items()}
    loc, vloc = m(**kw)
    _loss(loc, vloc, tgt.cuda()).backward()
    pp = {kk: v.detach().double().clone().requires_grad_(True) for kk, v in p.items()}
    ii = {kk: (v.double() if v.is_floating_point() else v) for kk, v in inp.items()}
    l, v = R.forward(pp, cfg, **ii)
    _loss(l, v, tgt.double()).backward()
    e_edge = max(rel_err(q.grad.cpu(), pp[n].grad) for n, q in m.named_parameters() if q.grad is not None and pp[n].grad is not None and ("edge_mlp." in n or "coord_mlp_r." in n))
    e_rest = max(rel_err(q.grad.cpu(), pp[n].grad) for n, q in m.named_parameters() if q.grad is not None and pp[n].grad is not None and not ("edge_mlp." in n or "coord_mlp_r." in n))
    E, N = inp["edge_index"].size(1), inp["node_loc"].size(0)
    print(f"N={N:4d} (N%16={N % 16:2d})  E={E:5d} (E%16={E % 16:2d})  loc err {rel_err(loc.cpu(), l.detach()):.2e}  edge-stage grads {e_edge:.2e}  other grads {e_rest:.2e}")
