"""one-off measurement of the WIDE path (hidden_nf = 128): fwd + loss + bwd on a Water-3D-like frame, eager launches.
usage: python tools/gpu_wide_timing.py [nodes] [channels] [hidden]"""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fastegnn_amd
from bench import make_frame, loss_fn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Hn = int(sys.argv[3]) if len(sys.argv) > 3 else 128
torch.manual_seed(43)
m = fastegnn_amd.FastEGNN(2, 0, 2, Hn, C, device="cuda", n_layers=4, gravity=[0, -1, 0])
frame, target = make_frame(n, C, 43, "cuda")
def step():
    for p in m.parameters(): p.grad = None
    loc, vloc = m(**frame)
    loss_fn(loc, vloc, target).backward()
step(); torch.cuda.synchronize()
t0 = time.time(); k = 3
for _ in range(k): step()
torch.cuda.synchronize()
ms = (time.time() - t0) / k * 1e3
E = frame["edge_index"].size(1)
print(f"wide path: hidden_nf={Hn} nodes={n} edges={E} C={C}: {ms:.1f} ms per step (fwd+loss+bwd, eager), peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
