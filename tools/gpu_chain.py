import ctypes as C, sys, torch
sys.path.insert(0, ".")  # run from the repository root
from fastegnn_amd import _lib as K
L = K.lib()
w = (torch.randn(4096, device="cuda") * 0.1)
out = torch.zeros(64 * 64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(iters, mode, waves, grid):
    L.fastegnn_selftest_chain(K.ptr(w), K.ptr(out), 10, mode, waves, grid, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); K.check(L.fastegnn_selftest_chain(K.ptr(w), K.ptr(out), iters, mode, waves, grid, st), "chain"); b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    fl = grid * waves * iters * 16 * 8192.0
    return fl / ms / 1e9
for mode, name in ((0, "gemm only, LDS image"), (1, "gemm+silu, LDS image"), (2, "gemm only, global image"), (3, "gemm+silu, global image")):
    for waves, grid in ((4, 256), (4, 512), (8, 256), (8, 512), (16, 256)):
        print(f"{name:26s} waves/WG {waves:2d} grid {grid:4d} ({waves*grid//1024} waves/SIMD): {run(2000, mode, waves, grid):7.1f} TFLOP/s")
