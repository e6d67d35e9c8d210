import ctypes as C, json, torch, sys
sys.path.insert(0, ".")  # run from the repository root
import fastegnn_amd
from fastegnn_amd import _lib as K
from bench import make_frame, loss_fn
L = K.lib()
torch.manual_seed(43)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
frame, target = make_frame(100000, 16, 43, "cuda")
def step(train=False):
    loc, vloc = m(**frame)
    if train: loss_fn(loc, vloc, target).backward()
out = (C.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(2): step()
    torch.cuda.synchronize(); L.fastegnn_debug_read_stamps(out, 1)
    for _ in range(3): step()
    torch.cuda.synchronize(); L.fastegnn_debug_read_stamps(out, 1)
v = list(out)[:8]; tot = sum(v)
names = ["idx+coords wait", "gather rows wait+pre", "silu", "gemm", "silu3+dot", "LDS transpose", "segment loop", "chunk bookkeeping"]
print("edge_fwd phase shares (forward-only run):")
for n, x in zip(names, v): print(f"  {n:24s} {x/tot*100:5.1f}%  ({x:.3g} cycles)")
