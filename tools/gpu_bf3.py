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
out2 = torch.zeros(16, 64, device="cuda")
K.check(L.fastegnn_selftest_chain_bf3(K.ptr(W), K.ptr(X), K.ptr(out2), 1, 4 | 8, 4, 1, st), "f16x2"); torch.cuda.synchronize()
print("f16x2  vs fp64: max err / max|y| = %.3e   rms %.3e (bf16x3 rms %.3e, torch fp32 rms %.3e)" % (
    (out2.cpu().double() - ref64).abs().max() / den, (out2.cpu().double() - ref64).pow(2).mean().sqrt() / den,
    (out.cpu().double() - ref64).pow(2).mean().sqrt() / den, (ref32.double() - ref64).pow(2).mean().sqrt() / den))
# the split itself, exactly: h = fp16(x) RNE, l = fp16((x - h) * 2048) -- against torch's conversions, on awkward magnitudes
for scale in (1.0, 1e-3, 37.0, 3e-5):
    Ws = W * scale
    K.check(L.fastegnn_selftest_chain_bf3(K.ptr(Ws), K.ptr(X), K.ptr(out2), 1, 4 | 8, 4, 1, st), "f16x2"); torch.cuda.synchronize()
    r64 = X.double().cpu() @ Ws.double().cpu().T
    # emulation of the three products in fp64 from torch's own fp16 roundings
    def sp(t):
        h = t.half().float(); l = ((t - h) * 2048).half().float(); return h.double().cpu(), l.double().cpu()
    wh, wl = sp(Ws); xh, xl = sp(X)
    emu = xh @ wh.T + (xh @ wl.T + xl @ wh.T) / 2048
    print("  scale %-7g f16x2 err %.3e of max|y|; against its fp64 emulation from torch's fp16 roundings %.3e" % (
        scale, (out2.cpu().double() - r64).abs().max() / r64.abs().max(), (out2.cpu().double() - emu).abs().max() / r64.abs().max()))
print("fp32 MFMA vs fp64:                %.3e" % ((Y.cpu().double() - ref64).abs().max() / den))
print("torch fp32 CPU vs fp64:           %.3e" % ((ref32.double() - ref64).abs().max() / den))
def run(iters, mode, waves, grid):
    L.fastegnn_selftest_chain_bf3(K.ptr(W), K.ptr(X), K.ptr(out), 10, mode, waves, grid, st); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); K.check(L.fastegnn_selftest_chain_bf3(K.ptr(W), K.ptr(X), K.ptr(out), iters, mode, waves, grid, st), "c"); b.record()
    torch.cuda.synchronize()
    return grid * waves * iters * 16 * 8192.0 / a.elapsed_time(b) / 1e9
for mode, name in ((0, "bf16x3 gemm+split"), (1, "bf16x3 gemm+split+silu"), (8, "f16x2 gemm+split"), (9, "f16x2 gemm+split+silu")):
    for waves, grid in ((4, 256), (8, 256), (8, 512), (16, 256)):
        print(f"{name:24s} waves/WG {waves:2d} grid {grid:4d} ({waves*grid//1024} waves/SIMD): {run(2000, mode, waves, grid):7.1f} fp32-equivalent TFLOP/s")
