"""diagnostic (-DFE_STAMP build, tools/gpu_stampbuild_vb2.sh): phase shares of virt_bwd_pc_kernel on the cfg4 frame"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import fastegnn_amd
from fastegnn_amd import _lib as K
from bench import make_frame, loss_fn
L = K.lib()
torch.manual_seed(43)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
NN = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
frame, target = make_frame(NN, 16, 43, "cuda")
def step():
    for p in m.parameters(): p.grad = None
    loc, vloc = m(**frame)
    loss_fn(loc, vloc, target).backward()
out = (C.c_ulonglong * 32)()
for _ in range(2): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_vb2_stamps(out, 1)
K.lib().fastegnn_profile_enable(1)
NS = 3
for _ in range(NS): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_vb2_stamps(out, 1)
prof = K.profile_collect()
v = list(out)
units = NS * 4 * ((NN + 15) // 16) * 16          # (tile, channel) pairs stamped
names = ["bookkeeping", "rows arrive, pre", "silu + split + V2", "silu, v store", "head x fwd + rank-1", "head X fwd + rank-1",
         "publish ring A", "Gv row + 2 transposed heads", "g_vp + publish ring B", "V2^T", "g_pre consumers", "final barrier wait"]
tot = sum(v[:12])
print("producer phases (cycles per (tile, channel), share):")
for n, x in zip(names, v[:12]): print(f"  {n:32s} {x / units:8.0f}  {x / tot * 100:5.1f}%")
print(f"  total {tot / units:.0f} cycles per (tile, channel) per producer wave")
for r, nm in enumerate(("X", "XX", "V2")):
    w, c = v[16 + 4 * r], v[17 + 4 * r]
    print(f"consumer {nm}: waiting {w / units:.0f}, reading+contracting {c / units:.0f}, loop {v[18 + 4 * r] / units:.0f} cycles per ticket")
print("consumers' final barrier wait per ticket:", v[12 + 11] / units if len(v) > 23 else 0)
for k in ("virt_bwd_kernel", "virt_bwd_gv_kernel", "virt_bwd_node_kernel"):
    print(k, "ms/launch (stamped):", prof[k][0] / prof[k][1])
