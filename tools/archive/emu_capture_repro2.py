import os, sys
os.environ["FASTEGNN_COMM"] = "abi"; os.environ.setdefault("FASTEGNN_SHARDED_SYNC", "0")
sys.path.insert(0, ".")
V = sys.argv[1]
import torch, fastegnn_amd, bench
from fastegnn_amd.sharded import ShardedFastEGNN
inp, _ = bench.make_frame(6000, 8, 5, "cuda", radius=0.035)
torch.manual_seed(3)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 8, device="cuda", n_layers=2, gravity=[0, -1, 0])
sm = ShardedFastEGNN(m, emulate=(4, 1))
local = sm.shard_inputs(**inp, reorder=True)
params = list(m.parameters())
def step():
    for p in params:
        p.grad = None
    loc, vloc = sm.forward_local(local)
    (loc.pow(2).mean() + vloc.pow(2).mean()).backward()
    return loc, vloc
if "a" in V:
    loc0, vloc0 = step()
else:
    step()
if "b" in V:
    g0 = [p.grad.clone() for p in params if p.grad is not None]
torch.cuda.synchronize()
gs = torch.cuda.Stream()
gs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(gs):
    step()
torch.cuda.current_stream().wait_stream(gs)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=gs):
    if "c" in V:
        loc1, vloc1 = step()
    else:
        step()
g.replay(); torch.cuda.synchronize()
print("OK", V)
