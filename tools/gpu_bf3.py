import ctypes as C, sys, torch
sys.path.insert(0, ".")  # run from the repository root
from fastegnn_amd import _lib as K
L = K.lib()
torch.manual_seed(0)
W = (torch.randn(64, 64) * 0.3).cuda(); X = torch.randn(16, 64).cuda()
out = torch.zeros(16, 64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
K.check(L.fastegnn_selftest_chain_bf3(K.ptr(W), K.ptr(X), K.ptr(out), 1, 4, 4, 1, st), "bf3")
torch.cuda.synchronize()
ref64 = (X.double().cpu() @ W.double().cpu().T)
ref32 = (X.cpu() @ W.cpu().T)
Y = torch.zeros(16, 64, device="cuda")
K.check(L.fastegnn_selftest_gemm(K.ptr(W), K.ptr(X), K.ptr(Y), 0, st), "gemm"); torch.cuda.synchronize()
den = ref64.abs().max()
print("bf16x3 vs fp64: max err / max|y| = %.3e" % ((out.cpu().double() - ref64).abs().max() / den))
print("fp32 MFMA vs fp64:                %.3e" % ((Y.cpu().double() - ref64).abs().max() / den))
print("torch fp32 CPU vs fp64:           %.3e" % ((ref32.double() - ref64).abs().max() / den))
def run(iters, mode, waves, grid):
    L.fastegnn_selftest_chain_bf3(K.ptr(W), K.ptr(X), K.ptr(out), 10, mode, waves, grid, st); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); K.check(L.fastegnn_selftest_chain_bf3(K.ptr(W), K.ptr(X), K.ptr(out), iters, mode, waves, grid, st), "c"); b.record()
    torch.cuda.synchronize()
    return grid * waves * iters * 16 * 8192.0 / a.elapsed_time(b) / 1e9
for mode, name in ((0, "bf16x3 gemm+split"), (1, "bf16x3 gemm+split+silu")):
    for waves, grid in ((4, 256), (8, 256), (8, 512), (16, 256)):
        print(f"{name:24s} waves/WG {waves:2d} grid {grid:4d} ({waves*grid//1024} waves/SIMD): {run(2000, mode, waves, grid):7.1f} fp32-equivalent TFLOP/s")
