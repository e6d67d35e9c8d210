"""Drop-in ``FastRF`` (reference ``models/FastRF.py``, built by ``main_*.py --model FastRF``): the FastEGNN
layer without node_model / node_model_virtual -- node and virtual features pass through every layer -- and
with the velocity scale computed from the detached norm of the velocity.  Same stage kernels as FastEGNN
(``FASTEGNN_F_RF``, include/fastegnn_hip.h); same constructor, ``forward`` and ``state_dict`` layout as the
reference class, whose name the harness dispatches on (``utils/train.py:57``)."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as K
from .model import FastEGNN


class E_GCL_vel(nn.Module):
    """Parameter holder of one FastRF layer; construction order follows models/FastRF.py:27-87."""

    def __init__(self, hidden_nf, node_attr_nf, edge_attr_nf, virtual_channels, act_fn, attention, tanh, gravity):
        super().__init__()
        Hn, Cn = hidden_nf, virtual_channels
        self.edge_mlp = nn.Sequential(nn.Linear(2 * Hn + 1 + edge_attr_nf, Hn), act_fn, nn.Linear(Hn, Hn), act_fn)
        self.edge_mlp_virtual = nn.Sequential(nn.Linear(2 * Hn + 1 + Cn, Hn), act_fn, nn.Linear(Hn, Hn), act_fn)
        if attention:
            self.att_mlp = nn.Sequential(nn.Linear(Hn, 1), nn.Sigmoid())
            self.att_mlp_virtual = nn.Sequential(nn.Linear(Hn, 1), nn.Sigmoid())

        def coord_mlp():
            last = nn.Linear(Hn, 1, bias=False)
            torch.nn.init.xavier_uniform_(last.weight, gain=0.001)
            mods = [nn.Linear(Hn, Hn), act_fn, last]
            if tanh:
                mods.append(nn.Tanh())
            return nn.Sequential(*mods)

        self.coord_mlp_r = coord_mlp()
        self.coord_mlp_r_virtual = coord_mlp()
        self.coord_mlp_v_virtual = coord_mlp()
        self.coord_mlp_vel = nn.Sequential(nn.Linear(1, Hn), act_fn, nn.Linear(Hn, 1))
        if gravity is not None:
            self.gravity_mlp = nn.Sequential(nn.Linear(Hn, Hn), act_fn, nn.Linear(Hn, 1))

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("E_GCL_vel is evaluated inside FastRF.forward by the HIP library")


class FastRF(FastEGNN):
    """MI355X-native drop-in for the reference ``FastRF`` (models/FastRF.py:189-238)."""

    _layer_cls = E_GCL_vel
    _extra_flags = K.F_RF

    def forward(self, node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=None, node_attr=None):
        # node_attr is accepted and ignored, as in the reference layer (it never reads it: FastRF.py:155-186)
        return super().forward(node_feat, node_loc, node_vel, edge_index, data_batch, loc_mean, edge_attr=edge_attr,
                               node_attr=None if self.node_attr_nf == 0 else node_attr)
