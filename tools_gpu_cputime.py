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
for _ in range(3): step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue (CPU) {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
