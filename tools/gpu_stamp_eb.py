"""diagnostic (-DFE_STAMP build, tools/gpu_stampbuild_vb.sh with this script): phase shares of the producers of edge_bwd_pc_kernel on the cfg4 frame"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import fastegnn_amd
from fastegnn_amd import _lib as K
from bench import make_frame, loss_fn
L = K.lib()
torch.manual_seed(43)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
frame, target = make_frame(100000, 16, 43, "cuda")
E = frame["edge_index"].size(1)
def step():
    for p in m.parameters(): p.grad = None
    loc, vloc = m(**frame)
    loss_fn(loc, vloc, target).backward()
out = (C.c_ulonglong * 16)()
for _ in range(2): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_eb_stamps(out, 1)
K.lib().fastegnn_profile_enable(1)
for _ in range(3): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_eb_stamps(out, 1)
prof = K.profile_collect()
v = list(out)[:11]; tot = sum(v); tiles = 3 * 4 * (E / 16.0)
names = ["idx + coordinates wait", "gathered rows wait + pre", "silu (x2)", "recompute products (x2)", "silu 3 + head dot", "degree / g_aggx rows, head adjoint, g_up",
         "publish (g_up, m)", "g_aggm row, WX1^T, att adjoint, g_mp", "publish (g_mp, t)", "W2^T, g_pre, g_d, per-edge stores", "transpose tile + row sums"]
print("edge_bwd producer phase shares (stamped build):")
for n, x in zip(names, v): print(f"  {n:44s} {x/tot*100:5.1f}%  ({x/tiles:.0f} cycles per tile)")
print("edge_bwd ms/launch (stamped):", prof["edge_bwd_kernel"][0] / prof["edge_bwd_kernel"][1])
