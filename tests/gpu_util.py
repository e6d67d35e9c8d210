"""Helpers shared by the -m gpu tests: build the HIP-backed module from a golden's weights."""
import torch

import fastegnn_amd
from tests.helpers import Golden


def act_module(c):
    """the nn.Module a reference user would pass as act_fn for this oracle Config"""
    from torch import nn
    q = c.act_param
    return {"silu": nn.SiLU(), "relu": nn.ReLU(), "leaky_relu": nn.LeakyReLU(q), "tanh": nn.Tanh(), "sigmoid": nn.Sigmoid(),
            "elu": nn.ELU(q if c.act == "elu" else 1.0), "gelu": nn.GELU(),
            "softplus": nn.Softplus(beta=q if c.act == "softplus" else 1.0)}[c.act]


def model_from_golden(g: Golden, device="cuda", cls=None):
    c = g.cfg
    m = (cls or fastegnn_amd.FastEGNN)(node_feat_nf=c.node_feat_nf, node_attr_nf=c.node_attr_nf, edge_attr_nf=c.edge_attr_nf,
                              hidden_nf=c.hidden_nf, virtual_channels=c.virtual_channels, device=device,
                              n_layers=c.n_layers, residual=c.residual, attention=c.attention,
                              normalize=c.normalize, tanh=c.tanh, gravity=c.gravity, act_fn=act_module(c))
    sd = {k: torch.from_numpy(v) for k, v in g.params.items()}
    m.load_state_dict(sd, strict=True)
    return m.to(device)
