"""diagnostic (-DFE_VBS_WATCHDOG build): one forward + backward through the channel-phased virtual backward; prints which spin gave up"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import fastegnn_amd
from fastegnn_amd import _lib as K
from bench import make_frame, loss_fn
L = K.lib()
torch.manual_seed(43)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=1, gravity=[0, -1, 0])
frame, target = make_frame(n, 16, 43, "cuda")
loc, vloc = m(**frame)
loss_fn(loc, vloc, target).backward()
torch.cuda.synchronize()
out = (C.c_int * 64)()
L.fastegnn_debug_read_vbs_dog(out, 1)
v = list(out)
names = {1: "wave3 waits filledA", 2: "producer waits READY", 3: "producer waits ring B drained", 4: "producer waits ring A drained[0] (wave 3)",
         5: "producer waits ring A drained[1] (wave 7)", 6: "wave7 loop"}
print("watchdog counts:", {names.get(i, i): v[i] for i in range(1, 8) if v[i]})
print("workgroup 0: wave7 doneA/doneB/total", v[38:41], "unit ctr", v[41], "headA", v[42], "headB", v[43])
print("  READY[0..8)", v[8:16], "DONE[0..8)", v[16:24])
print("  first give-up per wave (spin id * 1e6 + unit/ticket):", v[48:56])
g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
print("finite grads:", bool(torch.isfinite(g).all()), "loss ok")
