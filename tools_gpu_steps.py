import sys, time, torch
sys.path.insert(0, ".")
import fastegnn_amd
from bench import make_frame, loss_fn
torch.manual_seed(43)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
m.cache_graphs = False
frame, target = make_frame(100000, 16, 43, "cuda")
params = list(m.parameters())
def step():
    for p in params: p.grad = None
    loc, vloc = m(**frame)
    loss_fn(loc, vloc, target).backward()
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{t:.1f}" for t in ts))
print("mem GB", torch.cuda.max_memory_allocated() / 1e9, "reserved", torch.cuda.memory_reserved() / 1e9)
