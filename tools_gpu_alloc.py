"""diagnostic: device allocations and wall time per step at cfg5 size"""
import sys, time, torch
sys.path.insert(0, ".")
import fastegnn_amd
from bench import make_frame, loss_fn
N, C = int(sys.argv[1]), int(sys.argv[2])
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, C, device="cuda", n_layers=4, gravity=[0, -1, 0])
frame, target = make_frame(N, C, 43, "cuda")
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    s0 = torch.cuda.memory_stats()
    loc, vloc = m(**frame)
    torch.cuda.synchronize(); t1 = time.time()
    loss_fn(loc, vloc, target).backward()
    torch.cuda.synchronize(); t2 = time.time()
    s1 = torch.cuda.memory_stats()
    print(f"step {it}: fwd {1e3*(t1-t0):.1f} ms bwd {1e3*(t2-t1):.1f} ms  device mallocs {s1['num_device_alloc']-s0['num_device_alloc']} frees {s1['num_device_free']-s0['num_device_free']}  reserved {s1['reserved_bytes.all.current']/2**30:.1f} GiB peak alloc {s1['allocated_bytes.all.peak']/2**30:.1f} GiB")
