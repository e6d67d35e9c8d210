"""Helpers shared by the -m gpu tests: build the HIP-backed module from a golden's weights."""
import torch

import fastegnn_amd
from tests.helpers import Golden


def model_from_golden(g: Golden, device="cuda", cls=None):
    c = g.cfg
    m = (cls or fastegnn_amd.FastEGNN)(node_feat_nf=c.node_feat_nf, node_attr_nf=c.node_attr_nf, edge_attr_nf=c.edge_attr_nf,
                              hidden_nf=c.hidden_nf, virtual_channels=c.virtual_channels, device=device,
                              n_layers=c.n_layers, residual=c.residual, attention=c.attention,
                              normalize=c.normalize, tanh=c.tanh, gravity=c.gravity)
    sd = {k: torch.from_numpy(v) for k, v in g.params.items()}
    m.load_state_dict(sd, strict=True)
    return m.to(device)
