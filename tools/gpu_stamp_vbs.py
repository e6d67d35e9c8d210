"""diagnostic (-DFE_STAMP build, tools/gpu_r6_stamps.sh): phase shares of the PRODUCERS of virt_bwd_cs_kernel (the channel-phased
virtual backward, the default at cfg4) and of edge_bwd_pc_kernel on the cfg4 frame, in s_memtime ticks per unit and producer wave"""
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
E = frame["edge_index"].size(1)
def step():
    for p in m.parameters(): p.grad = None
    loc, vloc = m(**frame)
    loss_fn(loc, vloc, target).backward()
vb = (C.c_ulonglong * 32)()
eb = (C.c_ulonglong * 16)()
for _ in range(2): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_vb2_stamps(vb, 1); L.fastegnn_debug_read_eb_stamps(eb, 1)
K.lib().fastegnn_profile_enable(1)
NS = 3
for _ in range(NS): step()
torch.cuda.synchronize(); L.fastegnn_debug_read_vb2_stamps(vb, 1); L.fastegnn_debug_read_eb_stamps(eb, 1)
prof = K.profile_collect()
v = list(vb)
units = NS * 4 * ((NN + 15) // 16) * 16
names = ["ticket + phase / row waits", "head loads, pre, SiLU 1", "split + V2 product", "SiLU 2", "split v + 2 head products", "head x: SiLU, dot, rank-1, g_ux",
         "head X: the same", "g_np row + publish ring A", "3 grad splits + W3c^T + 2 transposed heads", "g_vp, publish ring B, V2^T", "g_pre: g_A RMW, pools, counters",
         "final barrier wait"]
tot = sum(v[:12])
print(f"virt_bwd_cs producers (N = {NN}): s_memtime ticks per (tile, channel) unit and producer wave, share")
for n, x in zip(names, v[:12]): print(f"  {n:46s} {x / units:8.1f}  {x / tot * 100:5.1f}%")
print(f"  total {tot / units:.1f} ticks per unit per producer wave")
e = list(eb)[:11]; etot = sum(e); tiles = NS * 4 * (E / 16.0)
enames = ["idx + coordinates wait", "gathered rows wait + pre", "silu (x2)", "recompute products (x2)", "silu 3 + head dot", "degree / g_aggx rows, head adjoint, g_up",
          "publish (g_up, m)", "g_aggm row, WX1^T, att adjoint, g_mp", "publish (g_mp, t)", "W2^T, g_pre, g_d, per-edge stores", "transpose tile + row sums"]
print("edge_bwd_pc producers: s_memtime ticks per 16-edge tile and producer wave, share")
for n, x in zip(enames, e): print(f"  {n:46s} {x / tiles:8.1f}  {x / etot * 100:5.1f}%")
print(f"  total {etot / tiles:.1f} ticks per tile per producer wave")
for k in ("virt_bwd_kernel", "edge_bwd_kernel"):
    print(k, "ms/launch (stamped):", round(prof[k][0] / prof[k][1], 4))
