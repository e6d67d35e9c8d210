"""diagnostic: do the parameter gradients share one flat storage after backward?"""
import sys, torch
sys.path.insert(0, ".")
import fastegnn_amd
from bench import make_frame, loss_fn
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
frame, target = make_frame(20000, 16, 43, "cuda")
loc, vloc = m(**frame); loss_fn(loc, vloc, target).backward()
ps = [p for p in m.parameters()]
st = {p.grad.untyped_storage().data_ptr() for p in ps if p.grad is not None}
print("params", len(ps), "with grad", sum(p.grad is not None for p in ps), "distinct storages", len(st))
print("storage bytes", {p.grad.untyped_storage().nbytes() for p in ps if p.grad is not None})
import time
from fastegnn_amd.dist import allreduce_gradients
