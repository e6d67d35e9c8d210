"""diagnostic (-DFE_STAMP build, tools/gpu_stampbuild_vb.sh): phase shares of virt_bwd_kernel on the cfg4 frame"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import fastegnn_amd
from fastegnn_amd import _lib as K
from bench import make_frame, loss_fn
L = K.lib()
torch.manual_seed(43)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
frame, target = make_frame(100000, 16, 43, "cuda")
def step():
    for p in m.parameters(): p.grad = None
    loc, vloc = m(**frame)
    loss_fn(loc, vloc, target).backward()
out = (C.c_ulonglong * 16)()
for _ in range(2): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_vb_stamps(out, 1)
K.lib().fastegnn_profile_enable(1)
for _ in range(3): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_vb_stamps(out, 1)
prof = K.profile_collect()
v = list(out)[:9]; tot = sum(v)
names = ["node-MLP adjoint (per tile)", "channel top: sync + W3cT stage + row requests", "pre, silu, V2, silu, operand", "W3cT product",
         "head x", "head X", "att adjoint, g_vp, V2T", "tile bookkeeping / tails", "g_pre consumers: g_A, vr, pools"]
print("virt_bwd phase shares (stamped build; wave 0 lane 0 of every wave):")
for n, x in zip(names, v): print(f"  {n:48s} {x/tot*100:5.1f}%  ({x/ (3*4*6250*16/1.0):.0f} cycles per (tile,channel))")
print("virt_bwd ms/launch (stamped):", prof["virt_bwd_kernel"][0] / prof["virt_bwd_kernel"][1])
