"""-m gpu: building blocks of the HIP library through the C ABI (MFMA image path, weight-gradient
GEMM, CSR build) against plain torch / the oracle's CSR."""
import ctypes as C

import pytest
import torch

from fastegnn_amd import _lib as K
from fastegnn_amd.model import SortedGraph
from oracle import factored as F

pytestmark = pytest.mark.gpu


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("transposed", [0, 1])
def test_mfma_image_gemm(transposed):
    g = torch.Generator().manual_seed(5)
    # asymmetric integer data: exact in fp32, catches any row/col or k-permutation slip
    W = torch.randint(-8, 9, (64, 64), generator=g).float().cuda()
    X = torch.randint(-8, 9, (16, 64), generator=g).float().cuda()
    Y = torch.zeros(16, 64, device="cuda")
    K.check(K.lib().fastegnn_selftest_gemm(K.ptr(W), K.ptr(X), K.ptr(Y), transposed, _st()), "selftest_gemm")
    A = W.T if transposed else W
    assert torch.equal(Y.cpu(), (X.cpu() @ A.cpu().T))


@pytest.mark.parametrize("transposed", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
def test_row_major_image_serves_product_and_transpose(transposed, mode):
    """One row-major bf16 image in LDS, read by rows (ds_read_b64) for W x and by columns (ds_read_b64_tr_b16) for W^T x.
    Integer data (exact in bf16 and in fp32): any row/column or k-order slip of the transposed read shows as a wrong
    integer; mode 0 additionally checks the three-part split on fp32 data against fp64."""
    g = torch.Generator().manual_seed(7 + transposed)
    W = torch.randint(-8, 9, (64, 64), generator=g).float().cuda()
    X = torch.randint(-8, 9, (16, 64), generator=g).float().cuda()
    Y = torch.zeros(16, 64, device="cuda")
    K.check(K.lib().fastegnn_selftest_rm(K.ptr(W), K.ptr(X), K.ptr(Y), transposed, mode, _st()), "selftest_rm")
    A = W.T if transposed else W
    assert torch.equal(Y.cpu(), (X.cpu() @ A.cpu().T))
    if mode == 0:
        Wf = torch.randn(64, 64, generator=g).cuda()
        Xf = torch.randn(16, 64, generator=g).cuda()
        K.check(K.lib().fastegnn_selftest_rm(K.ptr(Wf), K.ptr(Xf), K.ptr(Y), transposed, 0, _st()), "selftest_rm")
        Af = (Wf.T if transposed else Wf).double().cpu()
        ref = Xf.double().cpu() @ Af.T
        assert (Y.cpu().double() - ref).abs().max().item() < 4e-7 * ref.abs().max().item()


@pytest.mark.parametrize("M", [1, 15, 16, 257, 5000])
def test_wgrad_tn(M):
    g = torch.Generator().manual_seed(M)
    G = torch.randint(-4, 5, (M, 64), generator=g).float().cuda()
    T = torch.randint(-4, 5, (M, 64), generator=g).float().cuda()
    dW = torch.zeros(64, 64, device="cuda")
    db = torch.zeros(64, device="cuda")
    slab = torch.empty(K.lib().fastegnn_wg_slab_floats(), device="cuda")
    K.check(K.lib().fastegnn_selftest_wgrad(K.ptr(G), K.ptr(T), M, K.ptr(dW), K.ptr(db), K.ptr(slab), _st()),
            "selftest_wgrad")
    assert torch.equal(dW.cpu(), G.cpu().T @ T.cpu())
    assert torch.equal(db.cpu(), G.cpu().sum(0))


def _hub(N, E, hub_edges, g):
    """random edges plus one row AND one column that hold `hub_edges` edges each, scattered over the edge list"""
    ei = torch.randint(0, N, (2, E), generator=g)
    at = torch.randperm(E, generator=g)
    ei[0][at[:hub_edges]] = N // 3
    ei[1][at[hub_edges // 2: hub_edges // 2 + hub_edges]] = N // 2
    return ei


# (1000, 20000), (37, 400), (5, 3): <= 64 edges per id on average -- the index is built by COUNTING (csr.hip: count_index; one workgroup scans
# the bins); (40000, 590000): the same with rocprim's scan; (300, 70000): the radix sorts; "hub": ids beyond CS_BIG = 2048 edges inside a
# sparse graph (their edges are compacted by cs_big_kernel); the result is the stable sort's in every case
@pytest.mark.parametrize("N,E", [(1, 0), (5, 3), (37, 400), (1000, 20000), (300, 70000), (40000, 590000), ("hub", 0)])
def test_build_csr_matches_oracle(N, E):
    if N == "hub":
        N, E = 3000, 60000
        ei = _hub(N, E, 5000, torch.Generator().manual_seed(5))
    else:
        g = torch.Generator().manual_seed(N + E)
        ei = torch.randint(0, N, (2, E), generator=g)
    if N > 3:
        ei[0][ei[0] == 2] = 3     # node 2 has no in-edges
    ref = F.build_csr(ei, N)
    sg = SortedGraph(ei.cuda(), N)
    torch.cuda.synchronize()
    assert torch.equal(sg.rowptr.cpu().long(), ref.rowptr)
    assert torch.equal(sg.cscptr.cpu().long(), ref.cscptr)
    if E:
        assert torch.equal(sg.erow.cpu().long(), ref.row)
        assert torch.equal(sg.perm.cpu().long(), ref.perm)          # stable sort: identical order
        assert torch.equal(sg.col.cpu().long(), ref.col)
        assert torch.equal(sg.csc_eid.cpu().long(), ref.csc_eid)
    cr = sg.chunk_row.cpu()[: sg.n_chunks + 1]
    assert cr[0] == 0 and cr[-1] == N and bool((cr[1:] >= cr[:-1]).all())
    rp = sg.rowptr.cpu()
    T = K.lib().fastegnn_chunk_edges()
    for k in range(min(sg.n_chunks, 3000)):   # chunk k owns the rows whose first edge lies in [T k, T (k+1))
        for r in range(int(cr[k]), int(cr[k + 1])):
            assert T * k <= int(rp[r]) and (int(rp[r]) < T * (k + 1) or k == sg.n_chunks - 1)


def test_build_csr_counting_and_radix_forms_agree():
    """FASTEGNN_CSR_SORT forces one form (read once per process: child processes); a shard-shaped input (rows [row_begin, row_begin + n_rows)
    of a larger graph, a source table wider than the row range) and a dense one, both forms on both"""
    import subprocess, sys, os
    code = r'''
import sys, torch, hashlib
sys.path.insert(0, ".")
from fastegnn_amd.model import SortedGraph
g = torch.Generator().manual_seed(9)
out = []
for n_rows, n_src, row_begin, E in ((5000, 7000, 12000, 90000), (200, 200, 0, 30000)):
    ei = torch.stack([torch.randint(row_begin, row_begin + n_rows, (E,), generator=g), torch.randint(0, n_src, (E,), generator=g)])
    sg = SortedGraph(ei.cuda(), n_rows, n_src=n_src, row_begin=row_begin)
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for t in (sg.rowptr, sg.erow, sg.col, sg.perm, sg.cscptr, sg.csc_eid, sg.chunk_row[: sg.n_chunks + 1]):
        h.update(t.cpu().numpy().tobytes())
    out.append(h.hexdigest())
print("RESULT " + " ".join(out))
'''
    res = {}
    for form in ("count", "radix"):
        r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           env=dict(os.environ, FASTEGNN_CSR_SORT=form), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        res[form] = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    assert res["count"] == res["radix"], res


def test_bf16x3_split_gemm_is_fp32_accurate():
    """The 3-way bf16 split path (common.h: vsplit / gemm64_bf3) reproduces an fp32 GEMM to fp32
    rounding: error vs fp64 no larger than the exact-fp32 MFMA path's."""
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(64, 64, generator=g) * 0.3).cuda()
    X = torch.randn(16, 64, generator=g).cuda()
    out = torch.zeros(16, 64, device="cuda")
    ref = torch.zeros(16, 64, device="cuda")
    K.check(K.lib().fastegnn_selftest_chain_bf3(K.ptr(W), K.ptr(X), K.ptr(out), 1, 4, 4, 1, _st()), "chain_bf3")
    K.check(K.lib().fastegnn_selftest_gemm(K.ptr(W), K.ptr(X), K.ptr(ref), 0, _st()), "gemm")
    torch.cuda.synchronize()
    exact = X.double().cpu() @ W.double().cpu().T
    den = exact.abs().max()
    e_bf3 = (out.cpu().double() - exact).abs().max() / den
    e_f32 = (ref.cpu().double() - exact).abs().max() / den
    assert e_bf3 < 5e-7 and e_bf3 < 3 * e_f32 + 1e-7


def test_transposing_tile_sum_jreduce16():
    """common.h jreduce16 (DPP butterfly with bank masks, inline assembly): exact on integer data, every lane checked."""
    L = K.lib()
    g = torch.Generator().manual_seed(5)
    X = torch.randint(-1000, 1000, (16, 64), generator=g).float().cuda()
    out = torch.empty(64, device="cuda")
    K.check(L.fastegnn_selftest_jreduce(K.ptr(X), K.ptr(out), None), "selftest_jreduce")
    col = X.sum(0).cpu()
    want = torch.tensor([col[16 * (j >> 2) + 4 * q + (j & 3)] for q in range(4) for j in range(16)])
    assert torch.equal(out.cpu(), want)


def test_cross_lane_sums_of_a_wave():
    """common.h qsum (v_permlane16_swap / v_permlane32_swap, inline assembly) and jsum (DPP row rotations): exact on integer
    data, every lane checked -- the <v, w> of every fused kernel ends in them."""
    L = K.lib()
    g = torch.Generator().manual_seed(6)
    X = torch.randint(-100000, 100000, (64,), generator=g).float()
    out = torch.empty(128, device="cuda")
    K.check(L.fastegnn_selftest_lane_sums(K.ptr(X.cuda()), K.ptr(out), None), "selftest_lane_sums")
    q = X.view(4, 16).sum(0)                    # lanes l % 16 + 16 q
    j = X.view(4, 16).sum(1)                    # the 16 lanes of a row
    want = torch.cat([q.repeat(4), j.repeat_interleave(16)])
    assert torch.equal(out.cpu(), want)
