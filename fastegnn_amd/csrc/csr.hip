// Graph preprocessing: COO (int64, dataset order -- the reference's datasets emit edges sorted by
// length, datasets/*/dataset.py cutoff_edge) -> row-sorted CSR + col-keyed inverted index +
// edge-balanced row chunks.  Replaces the index plumbing of unsorted_segment_sum/mean
// (models/FastEGNN.py:279-294) with sorted segments so that the edge kernels reduce without atomics.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
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

// ---------------------------------------------------------------------------------------------------------------------------
// The same index by COUNTING (round 6): graphs whose ids have few edges each -- radius graphs, the shards of one -- do not need the radix
// sort's histogram + two or three onesweep passes (25 - 30 us each however few the keys: 114 us for the 240 k edges of a 1/8 shard of
// cfg4, 172 us for the whole frame).  count per id (atomics) -> exclusive scan -> place at ptr[id] + a per-id cursor (atomics: any order
// inside an id) -> every position finds its STABLE rank inside its id by counting the smaller original indices of its segment (d loads
// for an id of d edges, neighbours share the segment's lines).  The result is the stable sort's, element for element.  Ids beyond CS_BIG
// edges (no radius graph has them; a hub would cost d^2) are left out of the ranking and compacted in original order by a scan over
// all edges, one workgroup per such id.  Ids outside [0, n) land in an overflow bin behind the last id, as they sort behind it today.
constexpr int CS_BIG = 2048;
constexpr int CS_MAX_EDGES = 600000;   // beyond: the radix sorts (see fastegnn_build_csr)
constexpr int CS_SCAN1 = 32768;   // bins one workgroup scans by itself; beyond: rocprim's single-pass scan

template <bool ROWS>
__device__ inline int cs_key(const int64_t *ei, const int32_t *keys, int k, int row_begin, int n) {
  const long r = ROWS ? (long)(ei[k] - row_begin) : (long)keys[k];
  return (r < 0 || r >= n) ? n : (int)r;
}
template <bool ROWS>
__global__ void cs_count_kernel(const int64_t *ei, const int32_t *keys, int E, int row_begin, int n, int32_t *cnt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < E) atomicAdd(&cnt[cs_key<ROWS>(ei, keys, k, row_begin, n)], 1);
}
// ptr[i] = sum of cnt[0 .. i), i < m: one workgroup of 1024, 4096 bins per trip
__global__ __launch_bounds__(1024) void cs_scan_kernel(const int32_t *cnt, int m, int32_t *ptr) {
  __shared__ int part[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  for (int base = 0; base < m; base += 4096) {
    const int i = base + threadIdx.x * 4;
    int v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i + q < m ? cnt[i + q] : 0;
    const int s = v[0] + v[1] + v[2] + v[3];
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(inc, off);
      if (lane >= off) inc += t;
    }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int pw = part[w];
      before += w < wave ? pw : 0;
      total += pw;
    }
    int ex = carry + before + inc - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (i + q < m) ptr[i + q] = ex;
      ex += v[q];
    }
    carry += total;
    __syncthreads();
  }
}
// edge k -> an unordered place inside its id's segment; ptr -> the caller's rowptr / cscptr; ids beyond CS_BIG edges -> the list
// cs_big_kernel walks; (ROWS) the edge-balanced row chunks
template <bool ROWS>
__global__ void cs_place_kernel(const int64_t *ei, const int32_t *keys, int E, int row_begin, int n, const int32_t *ptr,
                                const int32_t *cnt, int32_t *cursor, int32_t *tmp_idx, int32_t *tmp_key, int32_t *ptr_out,
                                int32_t *n_big, int32_t *big_list, int n_chunks, int32_t *chunk_row) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < E) {
    const int r = cs_key<ROWS>(ei, keys, t, row_begin, n);
    const int pos = ptr[r] + atomicAdd(&cursor[r], 1);
    tmp_idx[pos] = t;
    tmp_key[pos] = r;
  }
  if (t <= n) {
    if (cnt[t] > CS_BIG) big_list[atomicAdd(n_big, 1)] = t;
    ptr_out[t] = ptr[t];
  }
  if (ROWS && t <= n_chunks) {
    int lo = n;
    if (t < n_chunks) {
      const int target = t * CHUNK_EDGES;
      int hi = n;
      lo = 0;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] < target) lo = mid + 1; else hi = mid;
      }
    }
    chunk_row[t] = lo;
  }
}
// position -> its stable rank inside the id's segment -> the outputs (ROWS: perm, erow, col; else: the col-keyed edge list)
template <bool ROWS>
__global__ void cs_emit_kernel(const int64_t *ei, int E, int n, const int32_t *ptr, const int32_t *cnt, const int32_t *tmp_idx,
                               const int32_t *tmp_key, int32_t *out_idx, int32_t *out_key, int32_t *out_col) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= E) return;
  const int r = tmp_key[pos];
  if (cnt[r] > CS_BIG) return;
  const int k = tmp_idx[pos], b = ptr[r], e = ptr[r + 1];
  int rank = 0;
  for (int j = b; j < e; ++j) rank += tmp_idx[j] < k ? 1 : 0;
  const int out = b + rank;
  out_idx[out] = k;
  if (ROWS) {
    out_key[out] = r;
    out_col[out] = (int32_t)ei[(size_t)E + k];
  }
}
// ids of more than CS_BIG edges: their edges in original order, by a pass over all edges (1024 per trip, block-wide prefix sums)
template <bool ROWS>
__global__ __launch_bounds__(256) void cs_big_kernel(const int64_t *ei, const int32_t *keys, int E, int row_begin, int n,
                                                    const int32_t *ptr, const int32_t *n_big, const int32_t *big_list,
                                                    int32_t *out_idx, int32_t *out_key, int32_t *out_col) {
  __shared__ int part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = *n_big;
  for (int i = blockIdx.x; i < nb; i += gridDim.x) {
    const int r = big_list[i];
    int run = ptr[r];
    for (int k0 = 0; k0 < E; k0 += 1024) {
      const int k = k0 + threadIdx.x * 4;
      bool f[4];
      int s = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f[q] = k + q < E && cs_key<ROWS>(ei, keys, k + q, row_begin, n) == r;
        s += f[q] ? 1 : 0;
      }
      int inc = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      if (lane == 63) part[wave] = inc;
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int pw = part[w];
        before += w < wave ? pw : 0;
        total += pw;
      }
      int out = run + before + inc - s;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!f[q]) continue;
        out_idx[out] = k + q;
        if (ROWS) {
          out_key[out] = r;
          out_col[out] = (int32_t)ei[(size_t)E + k + q];
        }
        ++out;
      }
      run += total;
      __syncthreads();
    }
  }
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// One id -> positions index by counting (see above).  ROWS: ids = edge_index[0] - row_begin, outputs ptr_out = rowptr, out_idx = perm,
// out_key = erow, out_col = col and the row chunks; else ids = keys[] (the row-sorted col array), outputs ptr_out = cscptr, out_idx = csc_eid.
// `ws`: (3 (n + 2) + 64 + E / CS_BIG + 2 + 2 E) ints + the scan's temporary beyond CS_SCAN1 bins
template <bool ROWS>
static int count_index(const int64_t *ei, const int32_t *keys, int E, int row_begin, int n, int32_t *ptr_out, int32_t *out_idx,
                       int32_t *out_key, int32_t *out_col, int n_chunks, int32_t *chunk_row, char *ws, size_t ws_bytes, hipStream_t st) {
  const int m = n + 2;   // n ids + the overflow bin + the total
  char *base = ws;
  auto take = [&](size_t ints) { int32_t *p = (int32_t *)base; base += align256(ints * 4); return p; };
  int32_t *cnt = take((size_t)2 * m + 64), *cursor = cnt + m, *n_big = cursor + m;   // one memset
  int32_t *ptr = take(m), *big_list = take((size_t)E / CS_BIG + 2), *tmp_idx = take(E), *tmp_key = take(E);
  FE_REQUIRE((size_t)(base - ws) <= ws_bytes, "build_csr: tmp too small for the counting index");
  (void)hipMemsetAsync(cnt, 0, ((size_t)2 * m + 64) * 4, st);
  const int ge = cdiv(E, 256);
  hipLaunchKernelGGL(cs_count_kernel<ROWS>, dim3(ge), dim3(256), 0, st, ei, keys, E, row_begin, n, cnt);
  if (m <= CS_SCAN1) {
    hipLaunchKernelGGL(cs_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, m, ptr);
  } else {
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, cnt, ptr, 0, (size_t)m, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) { set_error(std::string("build_csr: scan size query: ") + hipGetErrorString(e)); return FASTEGNN_E_LAUNCH; }
    FE_REQUIRE((size_t)(base - ws) + need <= ws_bytes, "build_csr: tmp too small for the scan");
    e = rocprim::exclusive_scan(base, need, cnt, ptr, 0, (size_t)m, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) { set_error(std::string("build_csr: scan: ") + hipGetErrorString(e)); return FASTEGNN_E_LAUNCH; }
  }
  int most = E > n + 1 ? E : n + 1;
  if (ROWS && n_chunks + 1 > most) most = n_chunks + 1;
  hipLaunchKernelGGL(cs_place_kernel<ROWS>, dim3(cdiv(most, 256)), dim3(256), 0, st, ei, keys, E, row_begin, n, ptr, cnt, cursor,
                     tmp_idx, tmp_key, ptr_out, n_big, big_list, n_chunks, chunk_row);
  hipLaunchKernelGGL(cs_emit_kernel<ROWS>, dim3(ge), dim3(256), 0, st, ei, E, n, ptr, cnt, tmp_idx, tmp_key, out_idx, out_key, out_col);
  hipLaunchKernelGGL(cs_big_kernel<ROWS>, dim3(64), dim3(256), 0, st, ei, keys, E, row_begin, n, ptr, n_big, big_list, out_idx, out_key,
                     out_col);
  return FASTEGNN_OK;
}

}  // namespace fe

using namespace fe;

extern "C" {

size_t fastegnn_chunk_rows(int32_t E) { return (size_t)(E / CHUNK_EDGES + 2); }
int32_t fastegnn_chunk_edges(void) { return CHUNK_EDGES; }

size_t fastegnn_csr_tmp_bytes(int32_t E, int32_t n_rows, int32_t n_src) {
  // keys_in, vals_in, keys_out(for the col sort) + radix-sort temporaries (double buffers + histograms); the counting index
  // (count_index) takes its 2 E + 3 (n + 2) ints from the same block
  const size_t n = (size_t)(n_rows > n_src ? n_rows : n_src) + 2;
  return 3 * align256((size_t)E * 4) + 4 * align256((size_t)E * 4) + 3 * align256(n * 4) + (8u << 20);
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
  // by counting while an id has <= 64 edges on average and the edge list is short: the counting form costs ~25 us + 0.13 us per 1 000 edges
  // (two atomics per edge), the radix sorts ~107 us + 0.034 -- measured inside the step on one MI355X: 55 against 115 us at the 240 k edges of
  // a 1/8 shard of cfg4, 274 against 172 us at the frame's 1.92 M (tools/gpu_r6_csr.sh).  FASTEGNN_CSR_SORT=radix | count forces one form
  static const char *force = getenv("FASTEGNN_CSR_SORT");
  bool counting = E > 0 && E <= CS_MAX_EDGES && (long)E <= 64l * n_rows && (long)E <= 64l * n_src;
  if (force && E > 0 && n_rows > 0 && n_src > 0) counting = strcmp(force, "count") == 0 ? true : strcmp(force, "radix") == 0 ? false : counting;
  if (counting) {
    int rc = count_index<true>(edge_index, nullptr, E, row_begin, n_rows, rowptr, perm, erow, col, nch, chunk_row, (char *)tmp, tmp_bytes, st);
    if (rc) return rc;
    if (want_csc && (rc = count_index<false>(nullptr, col, E, 0, n_src, cscptr, csc_eid, nullptr, nullptr, 0, nullptr, (char *)tmp, tmp_bytes, st)))
      return rc;
    return check_launch("build_csr");
  }
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
