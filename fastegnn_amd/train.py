"""Device-side training-step closure around the HIP FastEGNN.

Mirrors the body of the reference harness's mini-batch loop (``utils/train.py:30-170``): edge_attr
augmentation (:41-43), model call (:52), MSE + MMD loss (:104-165, with the sampled node indices passed
in explicitly instead of ``torch.randperm``), ``backward`` (:169) and Adam (``main_nbody.py:137``) --
each piece one C-ABI call of ``libfastegnn_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional

import torch

from . import _lib as K
from .model import _stream


def augment_edge_attr(edge_attr: Optional[torch.Tensor], loc_0: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """``cat([edge_attr, ||loc_0[row]-loc_0[col]||], 1)`` (utils/train.py:41-43)."""
    E = edge_index.size(1)
    k = edge_attr.size(1) if edge_attr is not None else 0
    out = torch.empty(E, k + 1, dtype=torch.float32, device=loc_0.device)
    ea = edge_attr.contiguous().float() if k else None
    K.check(K.lib().fastegnn_augment_edge_attr(K.ptr(edge_index.contiguous()), K.ptr(loc_0.contiguous().float()),
                                               K.ptr(ea), E, k, K.ptr(out), _stream(loc_0.device)),
            "fastegnn_augment_edge_attr")
    return out


class _MseMmd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc_pred, vloc, loc_t, sample_nodes, sigma, weight):
        dev = loc_pred.device
        loc_pred, vloc, loc_t = loc_pred.contiguous().float(), vloc.contiguous().float(), loc_t.contiguous().float()
        N, (B, _, Cn), S = loc_pred.size(0), vloc.shape, sample_nodes.size(1)
        loss2 = torch.empty(2, dtype=torch.float32, device=dev)
        g_loc, g_vloc = torch.empty_like(loc_pred), torch.empty_like(vloc)
        samp = sample_nodes.to(torch.int32).contiguous()
        K.check(K.lib().fastegnn_loss_mse_mmd(K.ptr(loc_pred), K.ptr(loc_t), K.ptr(vloc), K.ptr(samp), N, B, Cn, S,
                                              float(sigma), float(weight), K.ptr(loss2), K.ptr(g_loc), K.ptr(g_vloc),
                                              _stream(dev)), "fastegnn_loss_mse_mmd")
        ctx.save_for_backward(g_loc, g_vloc)
        ctx.mark_non_differentiable(loss2)
        return loss2[0], loss2

    @staticmethod
    def backward(ctx, g, _g2):
        g_loc, g_vloc = ctx.saved_tensors
        return g * g_loc, g * g_vloc, None, None, None, None


def mse_mmd_loss(loc_pred, vloc, loc_t, sample_nodes, sigma, weight):
    """-> (loss, mse): ``MSE(loc_pred, loc_t) + weight * (l_vv - l_rv)`` and the plain MSE the harness logs
    (utils/train.py:104-107,163-165).  ``sample_nodes`` [B,S]: absolute indices of the sampled real nodes."""
    loss, loss2 = _MseMmd.apply(loc_pred, vloc, loc_t, sample_nodes, sigma, weight)
    return loss, loss2[1]


class FusedAdam:
    """``torch.optim.Adam(params, lr, weight_decay)`` semantics (main_nbody.py:137) as multi-tensor HIP launches."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.step_count = 0
        n = len(self.params)
        self._numel = (C.c_int64 * n)(*[p.numel() for p in self.params])
        self._m = (C.c_void_p * n)(*[t.data_ptr() for t in self.exp_avg])
        self._v = (C.c_void_p * n)(*[t.data_ptr() for t in self.exp_avg_sq])

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        self.step_count += 1
        n = len(self.params)
        grads = []
        for p in self.params:
            g = p.grad
            if g is not None and (not g.is_contiguous() or g.dtype != torch.float32):
                g = g.contiguous().float()
            grads.append(g)
        gp = (C.c_void_p * n)(*[(g.data_ptr() if g is not None else None) for g in grads])
        dev = self.params[0].device
        # parameter pointers are read at every step: a model moved with .to() / .cuda() after the optimizer was built
        # has new storages (the moments follow the device of the parameters)
        pp = (C.c_void_p * n)(*[p.data_ptr() for p in self.params])
        for i, p in enumerate(self.params):
            if self.exp_avg[i].device != p.device:
                self.exp_avg[i], self.exp_avg_sq[i] = self.exp_avg[i].to(p.device), self.exp_avg_sq[i].to(p.device)
                self._m[i], self._v[i] = self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
        K.check(K.lib().fastegnn_adam_step(pp, gp, self._m, self._v, self._numel, n, self.step_count,
                                           float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                           float(self.eps), float(self.weight_decay), _stream(dev)),
                "fastegnn_adam_step")


def train_step(model, optimizer: FusedAdam, data: dict, sample_nodes, sigma, weight):
    """One iteration of utils/train.py:30-170 for the FastEGNN branch.  ``data`` holds the collated batch
    (loc_0, vel_0, loc_t, node_feat, edge_index, edge_attr, batch, loc_mean) on the GPU.  Returns (loss, mse)."""
    edge_attr = augment_edge_attr(data.get("edge_attr"), data["loc_0"], data["edge_index"])
    optimizer.zero_grad()
    loc_pred, vloc = model(node_loc=data["loc_0"], node_vel=data["vel_0"], node_attr=None,
                           node_feat=data["node_feat"], edge_index=data["edge_index"], loc_mean=data["loc_mean"],
                           data_batch=data["batch"], edge_attr=edge_attr)
    loss, mse = mse_mmd_loss(loc_pred, vloc, data["loc_t"], sample_nodes, sigma, weight)
    loss.backward()
    # the range guard of the f16x2 build (fastegnn_amd.model.RangeGuard), polled without synchronisation: when THIS poll finds that a
    # pass left the fp16 operand range the module moves to the wide-range build and the update is skipped (its gradients were zeroed
    # on the device anyway), like a skipped step of a loss scaler
    guard = getattr(model, "_range", None)
    if guard is not None and guard.poll(type(model).__name__, getattr(model, "_plist", None), why="a training step"):
        return loss.detach(), mse
    optimizer.step()
    return loss.detach(), mse
