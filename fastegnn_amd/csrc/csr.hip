// Graph preprocessing: COO (int64, dataset order -- the reference's datasets emit edges sorted by
// length, datasets/*/dataset.py cutoff_edge) -> row-sorted CSR + col-keyed inverted index +
// edge-balanced row chunks.  Replaces the index plumbing of unsorted_segment_sum/mean
// (models/FastEGNN.py:279-294) with sorted segments so that the edge kernels reduce without atomics.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "kernels.h"

namespace fe {

#ifndef FE_CHUNK_EDGES
#define FE_CHUNK_EDGES 32
#endif
constexpr int CHUNK_EDGES = FE_CHUNK_EDGES;   // the unit in which the edge kernels split rows among waves

__global__ void csr_keys_kernel(const int64_t *ei, int E, int row_begin, int32_t *keys, int32_t *vals) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  keys[k] = (int32_t)(ei[k] - row_begin);
  vals[k] = k;
}
__global__ void csr_col_kernel(const int64_t *ei, int E, const int32_t *perm, int32_t *col, int32_t *iota) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  col[k] = (int32_t)ei[(size_t)E + perm[k]];
  iota[k] = k;
}
// ptr[r] = first position in the ascending array `sorted` whose value is >= r, r in [0, n]
__global__ void lower_bound_kernel(const int32_t *sorted, int E, int n, int32_t *ptr) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  int lo = 0, hi = E;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (sorted[mid] < r) lo = mid + 1; else hi = mid;
  }
  ptr[r] = lo;
}
// chunk k owns the rows whose first edge position lies in [k*T, (k+1)*T)
__global__ void chunk_kernel(const int32_t *rowptr, int n_rows, int n_chunks, int32_t *chunk_row) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_chunks) return;
  if (k == n_chunks) { chunk_row[k] = n_rows; return; }
  int target = k * CHUNK_EDGES;
  int lo = 0, hi = n_rows;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (rowptr[mid] < target) lo = mid + 1; else hi = mid;
  }
  chunk_row[k] = lo;
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace fe

using namespace fe;

extern "C" {

size_t fastegnn_chunk_rows(int32_t E) { return (size_t)(E / CHUNK_EDGES + 2); }
int32_t fastegnn_chunk_edges(void) { return CHUNK_EDGES; }

size_t fastegnn_csr_tmp_bytes(int32_t E, int32_t n_rows, int32_t n_src) {
  (void)n_rows; (void)n_src;
  // keys_in, vals_in, keys_out(for the col sort) + radix-sort temporaries (double buffers + histograms)
  return 3 * align256((size_t)E * 4) + 4 * align256((size_t)E * 4) + (8u << 20);
}

int fastegnn_build_csr(const int64_t *edge_index, int32_t E, int32_t row_begin, int32_t n_rows, int32_t n_src,
                       int32_t *rowptr, int32_t *erow, int32_t *col, int32_t *perm, int32_t *cscptr,
                       int32_t *csc_eid, int32_t *chunk_row, int32_t *n_chunks, void *tmp, size_t tmp_bytes,
                       void *stream) {
  FE_REQUIRE(rowptr && chunk_row && n_chunks, "build_csr: null output");
  FE_REQUIRE(E == 0 || (edge_index && erow && col && perm && tmp), "build_csr: null pointer");
  FE_REQUIRE((cscptr == nullptr) == (csc_eid == nullptr), "build_csr: cscptr and csc_eid are given or omitted together");
  const bool want_csc = cscptr != nullptr;   // only the deterministic backward (FASTEGNN_F_DETERMINISTIC) reads the CSC index
  FE_REQUIRE(tmp_bytes >= fastegnn_csr_tmp_bytes(E, n_rows, n_src), "build_csr: tmp too small");
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps(K_CSR, st);
  const int nch = E / CHUNK_EDGES + 1;
  *n_chunks = nch;
  if (E > 0) {
    char *base = (char *)tmp;
    int32_t *keys_in = (int32_t *)base;
    base += align256((size_t)E * 4);
    int32_t *vals_in = (int32_t *)base;
    base += align256((size_t)E * 4);
    int32_t *keys_out = (int32_t *)base;
    base += align256((size_t)E * 4);
    void *rp_tmp = base;
    size_t rp_avail = tmp_bytes - (size_t)(base - (char *)tmp);
    size_t need = 0;
    // the keys are row / column ids: only the bits an id can have are sorted (17 at 100 000 nodes: three 8-bit passes of the
    // onesweep sort instead of four, 29 us each at cfg4)
    auto id_bits = [](int n) { int b = 1; while (b < 32 && (1ll << b) < (long long)n) ++b; return b; };
    const int row_bits = id_bits(n_rows), col_bits = id_bits(n_src);
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keys_in, erow, vals_in, perm, (size_t)E, 0, 32, st);
    if (e != hipSuccess) { set_error(std::string("build_csr: radix size query: ") + hipGetErrorString(e)); return FASTEGNN_E_LAUNCH; }
    FE_REQUIRE(need <= rp_avail, "build_csr: radix-sort temporary exceeds tmp buffer");
    const int g = cdiv(E, 256);
    hipLaunchKernelGGL(csr_keys_kernel, dim3(g), dim3(256), 0, st, edge_index, E, row_begin, keys_in, vals_in);
    e = rocprim::radix_sort_pairs(rp_tmp, need, keys_in, erow, vals_in, perm, (size_t)E, 0, row_bits, st);
    if (e != hipSuccess) { set_error(std::string("build_csr: row sort: ") + hipGetErrorString(e)); return FASTEGNN_E_LAUNCH; }
    hipLaunchKernelGGL(csr_col_kernel, dim3(g), dim3(256), 0, st, edge_index, E, perm, col, vals_in);
    hipLaunchKernelGGL(lower_bound_kernel, dim3(cdiv(n_rows + 1, 256)), dim3(256), 0, st, erow, E, n_rows, rowptr);
    if (want_csc) {
      e = rocprim::radix_sort_pairs(rp_tmp, need, col, keys_out, vals_in, csc_eid, (size_t)E, 0, col_bits, st);
      if (e != hipSuccess) { set_error(std::string("build_csr: col sort: ") + hipGetErrorString(e)); return FASTEGNN_E_LAUNCH; }
      hipLaunchKernelGGL(lower_bound_kernel, dim3(cdiv(n_src + 1, 256)), dim3(256), 0, st, keys_out, E, n_src, cscptr);
    }
  } else {
    (void)hipMemsetAsync(rowptr, 0, (size_t)(n_rows + 1) * 4, st);
    if (want_csc) (void)hipMemsetAsync(cscptr, 0, (size_t)(n_src + 1) * 4, st);
  }
  hipLaunchKernelGGL(chunk_kernel, dim3(cdiv(nch + 1, 256)), dim3(256), 0, st, rowptr, n_rows, nch, chunk_row);
  return check_launch("build_csr");
}

}  // extern "C"
