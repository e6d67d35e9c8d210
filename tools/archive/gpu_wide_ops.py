"""per-operator timing of the wide path's launches at cfg4's sizes (E = 1.92 M edge rows, N*C = 1.6 M virtual rows, H = 128):
time per call and the algorithmic HBM bytes per second.  usage: python tools/gpu_wide_ops.py [rows] [H]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fastegnn_amd import _lib as K

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1919172
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = K.lib()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
g = torch.Generator().manual_seed(1)
X = torch.randn(M, H, generator=g).cuda()
G = torch.randn(M, H, generator=g).cuda()
W = (torch.randn(H, 2 * H + 3, generator=g) / H ** 0.5).cuda()
b = torch.randn(H, generator=g).cuda()
out = torch.empty(M, H, device="cuda")
w1 = torch.randn(1, H, generator=g).cuda()
col1 = torch.empty(M, 1, device="cuda")
g1 = torch.randn(M, 1, generator=g).cuda()
Nn = M // 19
P = torch.randn(Nn, H, generator=g).cuda()
row = torch.sort(torch.randint(0, Nn, (M,), generator=g))[0].cuda()
col = torch.randint(0, Nn, (M,), generator=g).cuda()
dW = torch.zeros_like(W)
db = torch.zeros(H, device="cuda")
dw1 = torch.zeros_like(w1)
table = torch.zeros(Nn, H, device="cuda")
feat = torch.randn(M, 1, generator=g).cuda()
SILU, NONE = K.ACT_SILU, K.ACT_NONE
p = K.ptr
EH = M * H * 4


def t(name, fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:44s} {us:9.1f} us   {nbytes / us / 1e6:7.2f} TB/s (algorithmic)")


ck = K.check
t("linear  [M,H] x [H,H]", lambda: ck(L.fastegnn_wide_linear(p(X), M, H, p(W), W.size(1), 0, p(b), None, p(out), H, NONE, 0.0, st()), "l"), 2 * EH)
t("linear  act prologue", lambda: ck(L.fastegnn_wide_linear(p(X), M, H, p(W), W.size(1), 0, p(b), None, p(out), H, SILU, 0.0, st()), "l"), 2 * EH)
t("linear_dx", lambda: ck(L.fastegnn_wide_linear_dx(p(G), M, H, p(W), W.size(1), 0, H, p(out), 0, None, NONE, 0.0, st()), "l"), 2 * EH)
t("linear_dx * act'(Z)", lambda: ck(L.fastegnn_wide_linear_dx(p(G), M, H, p(W), W.size(1), 0, H, p(out), 0, p(X), SILU, 0.0, st()), "l"), 3 * EH)
t("linear_dw + db", lambda: ck(L.fastegnn_wide_linear_dw(p(G), p(X), M, H, H, p(dW), W.size(1), 0, p(db), NONE, 0.0, st()), "l"), 2 * EH)
t("linear_dw act prologue", lambda: ck(L.fastegnn_wide_linear_dw(p(G), p(X), M, H, H, p(dW), W.size(1), 0, p(db), SILU, 0.0, st()), "l"), 2 * EH)
t("linear  [M,H] x [H,1] act (head)", lambda: ck(L.fastegnn_wide_linear(p(X), M, H, p(w1), H, 0, None, None, p(col1), 1, SILU, 0.0, st()), "l"), EH)
t("linear_dx [M,1] x [1,H] * act'(Z) (head)", lambda: ck(L.fastegnn_wide_linear_dx(p(g1), M, 1, p(w1), H, 0, H, p(out), 0, p(X), SILU, 0.0, st()), "l"), 2 * EH)
t("linear_dw [1,H] act (head)", lambda: ck(L.fastegnn_wide_linear_dw(p(g1), p(X), M, 1, H, p(dw1), H, 0, None, SILU, 0.0, st()), "l"), EH)
t("linear  [M,1] x [1,H] (radial column)", lambda: ck(L.fastegnn_wide_linear(p(feat), M, 1, p(W), W.size(1), 2 * H, None, None, p(out), H, NONE, 0.0, st()), "l"), EH)
t("linear_dx [M,H] x [H,1] (radial column)", lambda: ck(L.fastegnn_wide_linear_dx(p(G), M, H, p(W), W.size(1), 2 * H, 1, p(col1), 0, None, NONE, 0.0, st()), "l"), EH)
t("linear_dw [H,1] (radial column)", lambda: ck(L.fastegnn_wide_linear_dw(p(G), p(feat), M, H, 1, p(dW), W.size(1), 2 * H, None, NONE, 0.0, st()), "l"), EH)
t("gather_add P[row] + base", lambda: ck(L.fastegnn_wide_gather_add(p(P), p(row), M, H, p(X), p(out), st()), "g"), 2 * EH)
t("gather_add P[col] + base", lambda: ck(L.fastegnn_wide_gather_add(p(P), p(col), M, H, p(X), p(out), st()), "g"), 2 * EH)
csort, cperm = torch.sort(col, stable=True)
Qn = torch.randn(Nn, H, generator=g).cuda()
t("gather2 P[row] + Q[col] + feat w", lambda: ck(L.fastegnn_wide_gather2(p(P), p(row), p(Qn), p(col), p(feat), 1, p(W), W.size(1), 2 * H, None, p(out), M, H, st()), "g2"), EH)
t("scatter_add_perm (col as sorted runs)", lambda: ck(L.fastegnn_wide_scatter_add_perm(p(table), p(csort), p(cperm), M, H, p(G), st()), "sp"), EH)
t("scatter_add sorted rows", lambda: ck(L.fastegnn_wide_scatter_add(p(table), p(row), M, H, p(G), st()), "s"), EH)
t("scatter_add unsorted rows", lambda: ck(L.fastegnn_wide_scatter_add(p(table), p(col), M, H, p(G), st()), "s"), EH)
t("act", lambda: ck(L.fastegnn_wide_act(p(X), M * H, SILU, 0.0, p(out), st()), "a"), 2 * EH)
t("act_backward", lambda: ck(L.fastegnn_wide_act_backward(p(X), p(G), M * H, SILU, 0.0, p(out), st()), "a"), 3 * EH)
