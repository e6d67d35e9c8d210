// Graph construction on device: the radius graph of the Water-3D dataset
// (datasets/simulation/dataset.py:80, torch_cluster radius_graph(r=0.035, no self loops)) and the
// "keep the shortest fraction" cutoff (datasets/*/dataset.py cutoff_edge) as a uniform-cell-list
// search: points are bucketed into cells of edge >= r, every point scans its 27 neighbouring cells.
// Two passes (count, fill) so that the caller allocates the exact edge list; edges come out grouped by
// centre node with neighbours in ascending index order (deterministic).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "kernels.h"

namespace fe {

struct Grid {
  float lo[3];
  float inv;      // 1 / cell edge
  int n[3];
};

// order-preserving map float -> unsigned (and back), so that atomicMin/atomicMax work on floats
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(256) void bbox_kernel(const float *loc, int N, unsigned *box /* [6]: min xyz, max xyz keys */) {
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
    for (int k = 0; k < 3; ++k) {
      const float v = loc[(size_t)i * 3 + k];
      mn[k] = fminf(mn[k], v);
      mx[k] = fmaxf(mx[k], v);
    }
  for (int k = 0; k < 3; ++k) {
    for (int off = 32; off > 0; off >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(box + k, f2key(mn[k]));
      atomicMax(box + 3 + k, f2key(mx[k]));
    }
  }
}

// grid from the bounding box: cell edge = max(r, extent / 255), and the per-axis cell count is clamped to 256, so
// that there are at most 256^3 = 2^24 cells (floor(extent / cell) + 1 <= 256; the clamp covers rounding)
__global__ void grid_kernel(const unsigned *box, float r, Grid *g) {
  if (threadIdx.x != 0) return;
  float cell = r;
  for (int k = 0; k < 3; ++k) {
    const float lo = key2f(box[k]), hi = key2f(box[3 + k]);
    g->lo[k] = lo;
    cell = fmaxf(cell, (hi - lo) / 255.0f);
  }
  g->inv = 1.0f / cell;
  for (int k = 0; k < 3; ++k) {
    const float lo = key2f(box[k]), hi = key2f(box[3 + k]);
    const int n = (int)((hi - lo) * g->inv) + 1;
    g->n[k] = n > 256 ? 256 : n;   // cell_of clamps the coordinates of the last cell accordingly
  }
}
__device__ __forceinline__ void cell_of(const Grid &g, const float *p, int c[3]) {
  for (int k = 0; k < 3; ++k) {
    int v = (int)((p[k] - g.lo[k]) * g.inv);
    c[k] = v < 0 ? 0 : (v >= g.n[k] ? g.n[k] - 1 : v);
  }
}
__global__ void cell_id_kernel(const float *loc, int N, const Grid *g, int32_t *cid, int32_t *idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c[3];
  cell_of(*g, loc + (size_t)i * 3, c);
  cid[i] = (c[2] * g->n[1] + c[1]) * g->n[0] + c[0];
  idx[i] = i;
}
// start[c] = first position in the sorted cell-id array with id >= c, for c in [0, ncell]
__global__ void cell_start_kernel(const int32_t *sorted_cid, int N, const Grid *g, int32_t *start, int max_cells) {
  const int ncell = g->n[0] * g->n[1] * g->n[2];
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ncell || c > max_cells) return;
  int lo = 0, hi = N;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (sorted_cid[mid] < c) lo = mid + 1; else hi = mid;
  }
  start[c] = lo;
}

// squared distance with every operation rounded separately (no fma contraction): bit-identical to the
// float32 reference arithmetic, so that the r-boundary decides identically
__device__ __forceinline__ float dist2(const float *a, const float *b) {
  const float dx = __fsub_rn(a[0], b[0]), dy = __fsub_rn(a[1], b[1]), dz = __fsub_rn(a[2], b[2]);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// FILL=false: deg[i] = number of j != i with |x_i - x_j|^2 <= r^2.
// FILL=true : writes the edges of centre i at offs[i].. with neighbours in ascending node order.
template <bool FILL>
__global__ __launch_bounds__(256) void radius_kernel(const float *loc, int N, const Grid *gp, const int32_t *sorted_idx,
                                                     const int32_t *start, float r2, int64_t *deg, const int64_t *offs,
                                                     int64_t *edge_index, float *dist, int64_t E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Grid g = *gp;
  const float pi[3] = {loc[(size_t)i * 3], loc[(size_t)i * 3 + 1], loc[(size_t)i * 3 + 2]};
  int c[3];
  cell_of(g, pi, c);
  int64_t cnt = 0;
  int64_t base = FILL ? offs[i] : 0;
  // neighbours are emitted in ascending node index: gather candidates of the 27 cells, insertion-sort
  // small runs locally (cells hold a handful of points)
  for (int dz = -1; dz <= 1; ++dz) {
    const int z = c[2] + dz;
    if (z < 0 || z >= g.n[2]) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = c[1] + dy;
      if (y < 0 || y >= g.n[1]) continue;
      const int x0 = max(c[0] - 1, 0), x1 = min(c[0] + 1, g.n[0] - 1);
      const int cell0 = (z * g.n[1] + y) * g.n[0] + x0;
      const int s = start[cell0], e = start[cell0 + (x1 - x0) + 1];   // the x-run of cells is contiguous
      for (int k = s; k < e; ++k) {
        const int j = sorted_idx[k];
        if (j == i) continue;
        const float pj[3] = {loc[(size_t)j * 3], loc[(size_t)j * 3 + 1], loc[(size_t)j * 3 + 2]};
        const float d2 = dist2(pi, pj);
        if (d2 <= r2) {
          if (FILL) {
            edge_index[base + cnt] = i;
            edge_index[E + base + cnt] = j;
            dist[base + cnt] = sqrtf(d2);
          }
          ++cnt;
        }
      }
    }
  }
  if (!FILL) deg[i] = cnt;
  if (FILL) {   // ascending neighbour order (insertion sort; degrees are tens)
    for (int64_t a = 1; a < cnt; ++a) {
      const int64_t cj = edge_index[E + base + a];
      const float cd = dist[base + a];
      int64_t b = a - 1;
      while (b >= 0 && edge_index[E + base + b] > cj) {
        edge_index[E + base + b + 1] = edge_index[E + base + b];
        dist[base + b + 1] = dist[base + b];
        --b;
      }
      edge_index[E + base + b + 1] = cj;
      dist[base + b + 1] = cd;
    }
  }
}

__global__ void iota_keys_kernel(const float *dist, int64_t E, uint32_t *keys, int32_t *vals) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  keys[k] = __float_as_uint(dist[k]);   // non-negative floats order like their bit patterns
  vals[k] = (int32_t)k;
}
__global__ void take_edges_kernel(const int64_t *ei, const float *dist, const int32_t *order, int64_t E, int64_t keep,
                                  int64_t *ei_out, float *dist_out) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= keep) return;
  const int32_t s = order[k];
  ei_out[k] = ei[s];
  ei_out[keep + k] = ei[E + s];
  if (dist_out) dist_out[k] = dist[s];
}

static size_t al(size_t x) { return (x + 255) / 256 * 256; }
struct RgWs {   // layout of the caller's workspace
  unsigned *box; Grid *grid; int32_t *cid, *idx, *cid_s, *idx_s, *start; int64_t *deg, *offs; void *tmp; size_t tmp_bytes;
};
constexpr int MAX_CELLS = 1 << 24;
static RgWs carve_ws(void *ws, size_t bytes, int N) {
  char *p = (char *)ws;
  RgWs w;
  w.box = (unsigned *)p; p += 256;
  w.grid = (Grid *)p; p += 256;
  w.cid = (int32_t *)p; p += al((size_t)N * 4);
  w.idx = (int32_t *)p; p += al((size_t)N * 4);
  w.cid_s = (int32_t *)p; p += al((size_t)N * 4);
  w.idx_s = (int32_t *)p; p += al((size_t)N * 4);
  w.start = (int32_t *)p; p += al((size_t)(MAX_CELLS + 2) * 4);
  w.deg = (int64_t *)p; p += al((size_t)(N + 1) * 8);
  w.offs = (int64_t *)p; p += al((size_t)(N + 1) * 8);
  w.tmp = p;
  w.tmp_bytes = bytes - (size_t)(p - (char *)ws);
  return w;
}

// ---------------------------------------------------------------- N-body: shortest ordered pairs of the complete graph
// One workgroup per system: the n(n-1) ordered pairs get the key (distance bits << 32 | pair index) -- distances are
// non-negative, so their bit patterns order like the values; the pair index makes the order total and deterministic --
// and are sorted by a bitonic network in LDS (n <= 128: 16 384 keys, 128 KB).  The first k keys are the edge list in
// ascending length (datasets/nbody/dataset.py:102-113 does this with cdist + topk on the host).
constexpr int NB_MAX_N = 128;
constexpr int NB_THREADS = 1024;
__global__ __launch_bounds__(NB_THREADS) void nbody_cutoff_kernel(const float *loc, int n, int k, int npad, int64_t *ei,
                                                                   float *ea) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long nb_keys[];
  __shared__ float xs[NB_MAX_N * 3];
  const int s = blockIdx.x, tid = threadIdx.x;
  const float *x = loc + (size_t)s * n * 3;
  for (int i = tid; i < n * 3; i += NB_THREADS) xs[i] = x[i];
  __syncthreads();
  for (int e = tid; e < npad; e += NB_THREADS) {
    unsigned long long key = ~0ull;
    if (e < n * n) {
      const int i = e / n, j = e - i * n;
      if (i != j) {
        const float dx = xs[3 * i] - xs[3 * j], dy = xs[3 * i + 1] - xs[3 * j + 1], dz = xs[3 * i + 2] - xs[3 * j + 2];
        const float d = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)e;
      }
    }
    nb_keys[e] = key;
  }
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < npad / 2; t += NB_THREADS) {
        const int lo = 2 * t - (t & (stride - 1));   // index with bit `stride` cleared
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = nb_keys[lo], b = nb_keys[hi];
        if ((a > b) == up) { nb_keys[lo] = b; nb_keys[hi] = a; }
      }
      __syncthreads();
    }
  for (int r = tid; r < k; r += NB_THREADS) {
    const unsigned long long key = nb_keys[r];
    const int e = (int)(unsigned)key, i = e / n, j = e - i * n;
    ei[((size_t)s * 2 + 0) * k + r] = i;
    ei[((size_t)s * 2 + 1) * k + r] = j;
    ea[(size_t)s * k + r] = __uint_as_float((unsigned)(key >> 32));
  }
}

}  // namespace fe

using namespace fe;

extern "C" {

size_t fastegnn_radius_graph_ws_bytes(int32_t N) {
  return 512 + 4 * al((size_t)N * 4) + al((size_t)(MAX_CELLS + 2) * 4) + 2 * al((size_t)(N + 1) * 8) +
         4 * al((size_t)N * 4) + (8u << 20);
}

int fastegnn_radius_graph_count(const float *loc, int32_t N, float r, void *ws, size_t ws_bytes, int64_t *n_edges,
                                void *stream) {
  FE_REQUIRE(loc && ws && n_edges && N >= 1 && r > 0.f, "radius_graph_count: bad argument");
  FE_REQUIRE(ws_bytes >= fastegnn_radius_graph_ws_bytes(N), "radius_graph_count: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps(K_MISC, st);
  RgWs w = carve_ws(ws, ws_bytes, N);
  (void)hipMemsetAsync(w.box, 0xff, 3 * sizeof(unsigned), st);       // minima keys: largest
  (void)hipMemsetAsync(w.box + 3, 0x00, 3 * sizeof(unsigned), st);   // maxima keys: smallest
  int g = cdiv(N, 256); if (g > 512) g = 512;
  hipLaunchKernelGGL(bbox_kernel, dim3(g), dim3(256), 0, st, loc, N, w.box);
  hipLaunchKernelGGL(grid_kernel, dim3(1), dim3(64), 0, st, w.box, r, w.grid);
  hipLaunchKernelGGL(cell_id_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, loc, N, w.grid, w.cid, w.idx);
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, w.cid, w.cid_s, w.idx, w.idx_s, (size_t)N, 0, 32, st);
  if (e != hipSuccess || need > w.tmp_bytes) { set_error("radius_graph_count: sort workspace"); return FASTEGNN_E_INVALID; }
  e = rocprim::radix_sort_pairs(w.tmp, need, w.cid, w.cid_s, w.idx, w.idx_s, (size_t)N, 0, 32, st);
  if (e != hipSuccess) { set_error("radius_graph_count: sort failed"); return FASTEGNN_E_LAUNCH; }
  hipLaunchKernelGGL(cell_start_kernel, dim3(cdiv(MAX_CELLS + 2, 256)), dim3(256), 0, st, w.cid_s, N, w.grid, w.start, MAX_CELLS + 1);
  hipLaunchKernelGGL(radius_kernel<false>, dim3(cdiv(N, 256)), dim3(256), 0, st, loc, N, w.grid, w.idx_s, w.start, r * r,
                     w.deg, (const int64_t *)nullptr, (int64_t *)nullptr, (float *)nullptr, (int64_t)0);
  // exclusive scan of the degrees (+ total at position N)
  (void)hipMemsetAsync(w.deg + N, 0, 8, st);
  need = 0;
  e = rocprim::exclusive_scan(nullptr, need, w.deg, w.offs, (int64_t)0, (size_t)N + 1, rocprim::plus<int64_t>(), st);
  if (e != hipSuccess || need > w.tmp_bytes) { set_error("radius_graph_count: scan workspace"); return FASTEGNN_E_INVALID; }
  e = rocprim::exclusive_scan(w.tmp, need, w.deg, w.offs, (int64_t)0, (size_t)N + 1, rocprim::plus<int64_t>(), st);
  if (e != hipSuccess) { set_error("radius_graph_count: scan failed"); return FASTEGNN_E_LAUNCH; }
  if (hipMemcpyAsync(n_edges, w.offs + N, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return check_launch("radius_graph_count(readback)");
  return check_launch("radius_graph_count");
}

int fastegnn_radius_graph_fill(const float *loc, int32_t N, float r, void *ws, size_t ws_bytes, int64_t n_edges,
                               int64_t *edge_index, float *dist, void *stream) {
  FE_REQUIRE(loc && ws && (n_edges == 0 || (edge_index && dist)), "radius_graph_fill: null pointer");
  if (n_edges == 0) return FASTEGNN_OK;
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps(K_MISC, st);
  RgWs w = carve_ws(ws, ws_bytes, N);
  hipLaunchKernelGGL(radius_kernel<true>, dim3(cdiv(N, 256)), dim3(256), 0, st, loc, N, w.grid, w.idx_s, w.start, r * r,
                     (int64_t *)nullptr, w.offs, edge_index, dist, n_edges);
  return check_launch("radius_kernel<fill>");
}

size_t fastegnn_cutoff_tmp_bytes(int64_t E) { return 4 * al((size_t)E * 4) + 4 * al((size_t)E * 4) + (8u << 20); }

int fastegnn_cutoff_edges(const int64_t *edge_index, const float *dist, int64_t E, int64_t keep, int64_t *edge_index_out,
                          float *dist_out, void *tmp, size_t tmp_bytes, void *stream) {
  FE_REQUIRE(keep >= 0 && keep <= E, "cutoff_edges: keep out of range");
  if (keep == 0) return FASTEGNN_OK;
  FE_REQUIRE(edge_index && dist && edge_index_out && tmp, "cutoff_edges: null pointer");
  FE_REQUIRE(tmp_bytes >= fastegnn_cutoff_tmp_bytes(E) && E < (1ll << 31), "cutoff_edges: tmp too small / E too large");
  hipStream_t st = (hipStream_t)stream;
  ProfScope _ps(K_MISC, st);
  char *p = (char *)tmp;
  uint32_t *keys = (uint32_t *)p; p += al((size_t)E * 4);
  int32_t *vals = (int32_t *)p; p += al((size_t)E * 4);
  uint32_t *keys_s = (uint32_t *)p; p += al((size_t)E * 4);
  int32_t *vals_s = (int32_t *)p; p += al((size_t)E * 4);
  const size_t avail = tmp_bytes - (size_t)(p - (char *)tmp);
  hipLaunchKernelGGL(iota_keys_kernel, dim3(cdiv(E, 256)), dim3(256), 0, st, dist, E, keys, vals);
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keys, keys_s, vals, vals_s, (size_t)E, 0, 32, st);
  if (e != hipSuccess || need > avail) { set_error("cutoff_edges: sort workspace"); return FASTEGNN_E_INVALID; }
  e = rocprim::radix_sort_pairs(p, need, keys, keys_s, vals, vals_s, (size_t)E, 0, 32, st);   // stable: ties keep edge order
  if (e != hipSuccess) { set_error("cutoff_edges: sort failed"); return FASTEGNN_E_LAUNCH; }
  hipLaunchKernelGGL(take_edges_kernel, dim3(cdiv(keep, 256)), dim3(256), 0, st, edge_index, dist, vals_s, E, keep,
                     edge_index_out, dist_out);
  return check_launch("cutoff_edges");
}

int fastegnn_nbody_cutoff_edges(const float *loc, int32_t S, int32_t n, int32_t k, int64_t *edge_index, float *dist,
                                void *stream) {
  FE_REQUIRE(S >= 0 && n >= 1 && k >= 0 && (int64_t)k <= (int64_t)n * (n - 1), "nbody_cutoff_edges: bad sizes");
  FE_REQUIRE(n <= NB_MAX_N, "nbody_cutoff_edges: more than 128 particles per system");
  if (S == 0 || k == 0) return FASTEGNN_OK;
  FE_REQUIRE(loc && edge_index && dist, "nbody_cutoff_edges: null pointer");
  int npad = 2;
  while (npad < n * n) npad <<= 1;
  const size_t lds = (size_t)npad * sizeof(unsigned long long);
  hipLaunchKernelGGL(nbody_cutoff_kernel, dim3(S), dim3(NB_THREADS), lds, (hipStream_t)stream, loc, n, k, npad, edge_index,
                     dist);
  return check_launch("nbody_cutoff_kernel");
}

}  // extern "C"
