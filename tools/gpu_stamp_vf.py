"""diagnostic: phase shares of virt_fwd_kernel (-DFE_STAMP_VF build; see tools/gpu_stampbuild_vf.sh)"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import fastegnn_amd
from fastegnn_amd import _lib as K
from bench import make_frame
L = K.lib()
L.fastegnn_debug_read_stamps_vf.argtypes = [C.c_void_p, C.c_int]
torch.manual_seed(43)
m = fastegnn_amd.FastEGNN(2, 0, 2, 64, 16, device="cuda", n_layers=4, gravity=[0, -1, 0])
frame, target = make_frame(100000, 16, 43, "cuda")
out = (C.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(2): m(**frame)
    torch.cuda.synchronize(); L.fastegnn_debug_read_stamps_vf(out, 1)
    n = 3
    for _ in range(n): m(**frame)
    torch.cuda.synchronize(); L.fastegnn_debug_read_stamps_vf(out, 1)
v = list(out)[:11]; tot = sum(v)
names = ["channel barriers + stage refill", "geometry + pre + silu 1", "split + product 1", "silu 2 + split", "products 2, 3 (heads)",
         "silu + head dots", "pools", "node-MLP block product", "tile head", "tile tail (node-level products)", "end"]
waves = 256 * 8 * 4 * n   # waves per launch x launches per forward x forwards
units = 6250 * 16 * 4 * n
print(f"virt_fwd phases (s_memtime ticks at 100 MHz; {tot / waves:.0f} ticks per wave per launch):")
for nm, x in zip(names, v): print(f"  {nm:34s} {x / tot * 100:5.1f}%   {x / units * 24:8.0f} cycles per (tile, channel) at 2.4 GHz")
