import ctypes as C, sys, torch
sys.path.insert(0, ".")
from fastegnn_amd import _lib as K
L = K.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
slab = torch.empty(L.fastegnn_wg_slab_floats(), device="cuda")
for M in (16, 1000, 100_000, 400_000, 1_600_000):
    G = torch.randn(M, 64, device="cuda"); T = torch.randn(M, 64, device="cuda")
    dW = torch.zeros(64, 64, device="cuda"); db = torch.zeros(64, device="cuda")
    ms = timed(lambda: K.check(L.fastegnn_selftest_wgrad(K.ptr(G), K.ptr(T), M, K.ptr(dW), K.ptr(db), K.ptr(slab), st), "wg"))
    print(f"wgrad tn+reduce M={M}: {ms*1e3:.1f} us  {2 * M * 256 / ms / 1e6:.0f} GB/s")
