"""repro harness: emulated rank + async ABI transport + HIP graph capture; argv: N layers world rank sync"""
import os, sys
N, Lr, W, R, SYNC = (int(x) for x in sys.argv[1:6])
os.environ["FASTEGNN_COMM"] = "abi"
os.environ["FASTEGNN_SHARDED_SYNC"] = str(SYNC)
sys.path.insert(0, ".")
import torch, fastegnn_amd, bench
from fastegnn_amd.sharded import ShardedFastEGNN
inp, _ = bench.make_frame(N, 8, 5, "cuda", radius=0.035)
torch.manual_seed(3)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 8, device="cuda", n_layers=Lr, gravity=[0, -1, 0])
sm = ShardedFastEGNN(m, emulate=(W, R))
local = sm.shard_inputs(**inp, reorder=True)
params = list(m.parameters())
def step():
    for p in params:
        p.grad = None
    loc, vloc = sm.forward_local(local)
    (loc.pow(2).mean() + vloc.pow(2).mean()).backward()
    return loc, vloc
for _ in range(int(os.environ.get("WARM", "1"))):
    step()
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
print("OK", sys.argv[1:])
