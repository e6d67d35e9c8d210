import os, sys
sys.path.insert(0, ".")
V = sys.argv[1]
import torch, fastegnn_amd, bench
from fastegnn_amd.sharded import ShardedFastEGNN
inp, _ = bench.make_frame(6000, 8, 5, "cuda", radius=0.035)
torch.manual_seed(3)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 8, device="cuda", n_layers=2, gravity=[0, -1, 0])
params = list(m.parameters())
if "s" in V:      # sharded path, world 1 (no emulation)
    sm = ShardedFastEGNN(m)
    local = sm.shard_inputs(**inp, reorder=True)
    fwd = lambda: sm.forward_local(local)
else:
    fwd = lambda: m(**inp)
def step():
    for p in params:
        p.grad = None
    loc, vloc = fwd()
    (loc.pow(2).mean() + vloc.pow(2).mean()).backward()
    return loc, vloc
keep = step()
if "d" in V:
    keep = (keep[0].detach().clone(), keep[1].detach().clone())
torch.cuda.synchronize()
gs = torch.cuda.Stream()
gs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(gs):
    step()
torch.cuda.current_stream().wait_stream(gs)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=gs):
    step()
g.replay(); torch.cuda.synchronize()
print("OK", V)
