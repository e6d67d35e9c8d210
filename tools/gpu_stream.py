"""diagnostic: achievable HBM streaming rates and the stand-alone rate of the weight-gradient contraction"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")  # run from the repository root
from fastegnn_amd import _lib as K
L = K.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
nf = 512 * 1024 * 1024   # 2 GiB
src = torch.ones(nf, device="cuda"); dst = torch.empty(nf, device="cuda")
for mode, name, mult in ((0, "read", 1), (1, "copy", 2), (2, "write", 1)):
    ms = timed(lambda: K.check(L.fastegnn_selftest_stream(K.ptr(src), K.ptr(dst), nf, mode, st), "stream"))
    print(f"stream {name:5s}: {mult * nf * 4 / ms / 1e6:8.1f} GB/s")
print("torch copy  :", 2 * nf * 4 / timed(lambda: dst.copy_(src)) / 1e6, "GB/s")
del src, dst
M = 2_000_000
G = torch.randn(M, 64, device="cuda"); T = torch.randn(M, 64, device="cuda")
dW = torch.zeros(64, 64, device="cuda"); db = torch.zeros(64, device="cuda")
slab = torch.empty(L.fastegnn_wg_slab_floats(), device="cuda")
ms = timed(lambda: K.check(L.fastegnn_selftest_wgrad(K.ptr(G), K.ptr(T), M, K.ptr(dW), K.ptr(db), K.ptr(slab), st), "wg"))
print(f"wgrad M={M}: {ms:.3f} ms  {2 * M * 256 / ms / 1e6:.0f} GB/s  {2 * M * 4096 / ms / 1e9:.1f} TFLOP/s")
dW.zero_(); L.fastegnn_selftest_wgrad(K.ptr(G), K.ptr(T), M, K.ptr(dW), K.ptr(db), K.ptr(slab), st); torch.cuda.synchronize()
ref = G.double().T @ T.double()
print("wgrad rel err vs fp64:", float((dW.double() - ref).abs().max() / ref.abs().max()))
