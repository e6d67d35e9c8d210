// Generic-width primitives of the WIDE path: FastEGNN with 64 < hidden_nf <= 256 (reference: any --dim_hidden,
// main_nbody.py:27, models/FastEGNN.py:28-99).  The fused stage kernels of this library are built on 64-wide register
// tiles; a wider model runs UNFUSED on the operators below -- the op sequence of models/FastEGNN.py:102-223 with every
// hidden-sized tensor op as one of these launches (fastegnn_amd/wide.py assembles them, autograd composes the backward from
// the *_dx / *_dw / *_bwd entry points).  Correctness first: plain fp32 arithmetic (v_mfma_f32_16x16x4_f32 in the two
// LDS-tiled 64x64 GEMM kernels, vector FMAs elsewhere; no operand splits), fp32 atomics for the row-keyed sums.
// DESIGN.md section 9 prices this path; the tuned path is hidden_nf <= 64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef FE_ACT_GENERIC
#define FE_ACT_GENERIC   // this translation unit always carries every activation kind (act_both)
#endif
#include "kernels.h"

namespace fe {
namespace wide {

constexpr int BM = 64, BN = 64, BK = 16;

// C[m, n] = (base ? base[m*ldc + n] : 0) + (bias ? bias[n] : 0) + sum_k A[m*lda + k] * B(k, n),   B(k, n) = Bp[k*sbk + n*sbn]
// (forward of a Linear: B(k, o) = W[o*ldw + c0 + k]; its input gradient: B(o, k) = W[o*ldw + c0 + k])
// 256 threads = 4 waves, one 64 x 64 tile per workgroup, operands staged through LDS, fp32 products on the matrix pipe.
__global__ __launch_bounds__(256) void gemm_tile_kernel(const float *A, int lda, long M, int Kd, const float *Bp, long sbk, long sbn,
                                                        int N, const float *bias, const float *base, float *C, int ldc,
                                                        int accumulate) {
  __shared__ float As[BK][BM + 16];
  __shared__ float Bs[BK][BN + 16];
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int tid = threadIdx.x;
  // wave w owns the 32 x 32 quadrant (w >> 1, w & 1) of the tile as 2 x 2 blocks of v_mfma_f32_16x16x4_f32 (fp32 products):
  // lane (lk, li) supplies A[row li][k lk] and B[k lk][column li] of a block and holds rows 4 lk .. 4 lk + 3 of column li of D
  const int wave = tid >> 6, li = tid & 15, lk = (tid >> 4) & 3;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < Kd; k0 += BK) {
    // A tile: 64 rows x 16 k -- thread loads rows (tid >> 4) + 16 i, k = tid & 15 (consecutive k of a row: coalesced)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 4) + 16 * i, k = tid & 15;
      const long m = m0 + r;
      As[k][r] = (m < M && k0 + k < Kd) ? A[(size_t)m * lda + k0 + k] : 0.f;
    }
    // B tile: 16 k x 64 n.  Walk the faster-varying index of B with the fast thread index.
    if (sbn == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = (tid >> 6) + 4 * i, n = tid & 63;
        Bs[k][n] = (k0 + k < Kd && n0 + n < N) ? Bp[(size_t)(k0 + k) * sbk + (size_t)(n0 + n)] : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = (tid >> 4) + 16 * i, k = tid & 15;
        Bs[k][n] = (k0 + k < Kd && n0 + n < N) ? Bp[(size_t)(k0 + k) * sbk + (size_t)(n0 + n) * sbn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      const float a0 = As[ks + lk][wr + li], a1 = As[ks + lk][wr + 16 + li];
      const float b0 = Bs[ks + lk][wc + li], b1 = Bs[ks + lk][wc + 16 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + wr + 16 * bi + 4 * lk + r;
      if (m >= M) continue;
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int n = n0 + wc + 16 * bj + li;
        if (n >= N) continue;
        float v = acc[bi][bj][r];
        if (bias) v += bias[n];
        if (base) v += base[(size_t)m * ldc + n];
        float *dst = C + (size_t)m * ldc + n;
        *dst = accumulate ? *dst + v : v;
      }
    }
}

// the same product for N <= 8 columns (the [1, H] heads and their input gradients, the rank-1 radial columns): sixteen lanes per
// row, each summing every 16th k, combined with a butterfly
__global__ __launch_bounds__(256) void gemm_smalln_kernel(const float *A, int lda, long M, int Kd, const float *Bp, long sbk, long sbn,
                                                          int N, const float *bias, const float *base, float *C, int ldc,
                                                          int accumulate) {
  const long m = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const bool live = m < M;
  for (int n = 0; n < N; ++n) {
    float s = 0.f;
    if (live)
      for (int k = sub; k < Kd; k += 16) s += A[(size_t)m * lda + k] * Bp[(size_t)k * sbk + (size_t)n * sbn];
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 16);
    if (live && sub == 0) {
      float v = s;
      if (bias) v += bias[n];
      if (base) v += base[(size_t)m * ldc + n];
      float *dst = C + (size_t)m * ldc + n;
      *dst = accumulate ? *dst + v : v;
    }
  }
}

// dW[o*ldw + c0 + k] += sum_m G[m*ldg + o] * X[m*ldx + k]  over the workgroup's row range (blockIdx.z), fp32 atomics into dW
__global__ __launch_bounds__(256) void tn_tile_kernel(const float *G, int ldg, const float *X, int ldx, long M, int O, int Kd,
                                                      float *dW, int ldw, int c0, long rows_per_split, float *db) {
  __shared__ float Gs[BK][BM + 16];   // [row][o]
  __shared__ float Xs[BK][BN + 16];   // [row][k]
  const int o0 = blockIdx.x * BM, k0 = blockIdx.y * BN;
  const long r_lo = (long)blockIdx.z * rows_per_split, r_hi = r_lo + rows_per_split < M ? r_lo + rows_per_split : M;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, li = tid & 15, lk = (tid >> 4) & 3;   // as gemm_tile_kernel, the contraction runs over the rows
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // db (bias gradient, may be null): the column sums of G ride along in the workgroups of the first k-block -- thread c < 64 adds
  // the 16 rows of its column from the staged tile (in double: cancelling sums over up to 10^6 rows)
  const bool do_bias = db != nullptr && blockIdx.y == 0 && tid < BM;
  double bsum = 0.0;
  for (long r0 = r_lo; r0 < r_hi; r0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 6) + 4 * i, c = tid & 63;
      const long m = r0 + r;
      Gs[r][c] = (m < r_hi && o0 + c < O) ? G[(size_t)m * ldg + o0 + c] : 0.f;
      Xs[r][c] = (m < r_hi && k0 + c < Kd) ? X[(size_t)m * ldx + k0 + c] : 0.f;
    }
    __syncthreads();
    if (do_bias) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < BK; ++r) t += Gs[r][tid];
      bsum += (double)t;
    }
#pragma unroll
    for (int rs = 0; rs < BK; rs += 4) {
      const float a0 = Gs[rs + lk][wr + li], a1 = Gs[rs + lk][wr + 16 + li];
      const float b0 = Xs[rs + lk][wc + li], b1 = Xs[rs + lk][wc + 16 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = o0 + wr + 16 * bi + 4 * lk + r;
      if (o >= O) continue;
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int k = k0 + wc + 16 * bj + li;
        if (k < Kd) atomicAdd(dW + (size_t)o * ldw + c0 + k, acc[bi][bj][r]);
      }
    }
  if (do_bias && o0 + tid < O) atomicAdd(db + o0 + tid, (float)bsum);
}

// out[s*so + l*sl] += sum_m S[m*lds + s] * L[m*ldl + l]   with a SMALL side (ns <= 8 columns): the [1, H] heads' weight gradients
// (S = G, L = X) and the few feature columns of a first layer (S = X, L = G).  Thread = one column l, rows of the workgroup's range.
__global__ __launch_bounds__(256) void tn_small_kernel(const float *S, int lds_, int ns, const float *L, int ldl, int nl, long M,
                                                       float *out, long so, long sl, long rows_per_split) {
  const long r_lo = (long)blockIdx.y * rows_per_split, r_hi = r_lo + rows_per_split < M ? r_lo + rows_per_split : M;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= nl) return;
  double acc[8] = {};   // (cancelling column sums over up to 10^6 rows: the range's partial in double, one fp32 atomic per range)
  for (long m = r_lo; m < r_hi; ++m) {
    const double x = L[(size_t)m * ldl + l];
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < ns) acc[s] += (double)S[(size_t)m * lds_ + s] * x;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s)
    if (s < ns) atomicAdd(out + (size_t)s * so + (size_t)l * sl, (float)acc[s]);
}

// db[o] += sum_m G[m*ldg + o]
__global__ __launch_bounds__(256) void colsum_kernel(const float *G, int ldg, long M, int O, float *db, long rows_per_split) {
  const long r_lo = (long)blockIdx.y * rows_per_split, r_hi = r_lo + rows_per_split < M ? r_lo + rows_per_split : M;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= O) return;
  double s = 0.0;
  for (long m = r_lo; m < r_hi; ++m) s += (double)G[(size_t)m * ldg + o];
  atomicAdd(db + o, (float)s);
}

__global__ __launch_bounds__(256) void act_kernel(const float *z, size_t n, Act a, float *y) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = act_f(z[i], a);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *z, const float *dy, size_t n, Act a, float *dz) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dz[i] = dy[i] * dact_f(z[i], a);
}

// out[m, :] = (base ? base[m, :] : 0) + X[idx[m], :]
__global__ __launch_bounds__(256) void gather_add_kernel(const float *X, const int64_t *idx, long M, int W, const float *base, float *out) {
  const size_t n = (size_t)M * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t m = i / W, w = i - m * W;
    const float v = X[(size_t)idx[m] * W + w];
    out[i] = base ? base[i] + v : v;
  }
}
// table[idx[m], :] += rows[m, :]
__global__ __launch_bounds__(256) void scatter_add_kernel(float *table, const int64_t *idx, long M, int W, const float *rows) {
  const size_t n = (size_t)M * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t m = i / W, w = i - m * W;
    atomicAdd(table + (size_t)idx[m] * W + w, rows[i]);
  }
}
// the same for wide rows (W >= 32): thread = one column, workgroup = a range of rows walked in order with a running sum that is
// flushed (one atomic) whenever the target changes -- sorted or grouped indices (segment sums by graph, the C channel rows of a
// node) cost one atomic per run instead of one per row; unsorted ones cost what scatter_add_kernel costs
__global__ __launch_bounds__(256) void scatter_add_runs_kernel(float *table, const int64_t *idx, long M, int W, const float *rows,
                                                               long rows_per_wg) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const long m_lo = (long)blockIdx.y * rows_per_wg, m_hi = m_lo + rows_per_wg < M ? m_lo + rows_per_wg : M;
  if (w >= W || m_lo >= m_hi) return;
  int64_t cur = idx[m_lo];
  float acc = 0.f;
  for (long m = m_lo; m < m_hi; ++m) {
    const int64_t i = idx[m];
    if (i != cur) {
      atomicAdd(table + (size_t)cur * W + w, acc);
      acc = 0.f;
      cur = i;
    }
    acc += rows[(size_t)m * W + w];
  }
  atomicAdd(table + (size_t)cur * W + w, acc);
}
// Y[m, :] = X[m, :] * s[m]
__global__ __launch_bounds__(256) void rowscale_kernel(const float *X, const float *s, long M, int W, float *Y) {
  const size_t n = (size_t)M * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) Y[i] = X[i] * s[i / W];
}
// out[m] = sum_w A[m, w] * B[m, w]
__global__ __launch_bounds__(256) void rowdot_kernel(const float *A, const float *B, long M, int W, float *out) {
  const long m = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  float s = 0.f;
  if (m < M)
    for (int w = sub; w < W; w += 16) s += A[(size_t)m * W + w] * B[(size_t)m * W + w];
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 16);
  if (m < M && sub == 0) out[m] = s;
}

inline int grid1d(size_t n) {
  const size_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}
// row ranges of the reductions over M: enough workgroups to fill the chip, at least 512 rows each
inline int row_splits(long M, long other_wgs) {
  long want = (2048 + other_wgs - 1) / (other_wgs > 0 ? other_wgs : 1);
  long most = (M + 511) / 512;
  if (want > most) want = most;
  if (want < 1) want = 1;
  if (want > 4096) want = 4096;
  return (int)want;
}

static int gemm(const float *A, int lda, long M, int Kd, const float *Bp, long sbk, long sbn, int N, const float *bias,
                const float *base, float *C, int ldc, int accumulate, hipStream_t st, const char *what) {
  if (M == 0 || N == 0) return FASTEGNN_OK;
  if (N <= 8) {
    hipLaunchKernelGGL(gemm_smalln_kernel, dim3((unsigned)cdiv(M, 16)), dim3(256), 0, st, A, lda, M, Kd, Bp, sbk, sbn, N, bias, base, C,
                       ldc, accumulate);
  } else {
    hipLaunchKernelGGL(gemm_tile_kernel, dim3((unsigned)cdiv(M, BM), (unsigned)cdiv(N, BN)), dim3(256), 0, st, A, lda, M, Kd, Bp, sbk,
                       sbn, N, bias, base, C, ldc, accumulate);
  }
  return check_launch(what);
}

}  // namespace wide
}  // namespace fe

using namespace fe;
using namespace fe::wide;

extern "C" {

// out[M, O] = (base ? base : 0) + X[M, K] . W[:, c0 : c0 + K]^T + bias        (models/FastEGNN.py: every nn.Linear; a Linear over a
// torch.cat of inputs is the sum of these calls over the weight's column blocks, chained through `base`)
int fastegnn_wide_linear(const float *X, int64_t M, int32_t K, const float *W, int32_t ldw, int32_t c0, const float *bias,
                         const float *base, float *out, int32_t O, void *stream) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0, "fastegnn_wide_linear: bad sizes");
  FE_REQUIRE((X || M == 0) && W && (out || M == 0), "fastegnn_wide_linear: null pointer");
  return gemm(X, K, M, K, W + c0, 1, ldw, O, bias, base, out, O, 0, (hipStream_t)stream, "fastegnn_wide_linear");
}
// dX[M, K] (+)= G[M, O] . W[:, c0 : c0 + K]
int fastegnn_wide_linear_dx(const float *G, int64_t M, int32_t O, const float *W, int32_t ldw, int32_t c0, int32_t K, float *dX,
                            int32_t accumulate, void *stream) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0, "fastegnn_wide_linear_dx: bad sizes");
  FE_REQUIRE((G || M == 0) && W && (dX || M == 0), "fastegnn_wide_linear_dx: null pointer");
  return gemm(G, O, M, O, W + c0, ldw, 1, K, nullptr, nullptr, dX, K, accumulate, (hipStream_t)stream, "fastegnn_wide_linear_dx");
}
// dW[:, c0 : c0 + K] += G[M, O]^T . X[M, K];  db[O] += column sums of G (db may be null)
int fastegnn_wide_linear_dw(const float *G, const float *X, int64_t M, int32_t O, int32_t K, float *dW, int32_t ldw, int32_t c0,
                            float *db, void *stream) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0, "fastegnn_wide_linear_dw: bad sizes");
  FE_REQUIRE((G && X) || M == 0, "fastegnn_wide_linear_dw: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) return FASTEGNN_OK;
  if (dW) {
    if (O <= 8) {          // small side = G's columns, long side = X's
      const int gx = cdiv(K, 256), ns = row_splits(M, gx);
      hipLaunchKernelGGL(tn_small_kernel, dim3(gx, ns), dim3(256), 0, st, G, O, O, X, K, K, (long)M, dW + c0, (long)ldw, (long)1,
                         (long)cdiv(M, ns));
    } else if (K <= 8) {   // small side = X's columns
      const int gx = cdiv(O, 256), ns = row_splits(M, gx);
      hipLaunchKernelGGL(tn_small_kernel, dim3(gx, ns), dim3(256), 0, st, X, K, K, G, O, O, (long)M, dW + c0, (long)1, (long)ldw,
                         (long)cdiv(M, ns));
    } else {
      const int gx = cdiv(O, BM), gy = cdiv(K, BN), ns = row_splits(M, (long)gx * gy);
      long rows = cdiv(M, ns);
      rows = (rows + BK - 1) / BK * BK;
      hipLaunchKernelGGL(tn_tile_kernel, dim3(gx, gy, (unsigned)cdiv(M, rows)), dim3(256), 0, st, G, O, X, K, (long)M, O, K, dW, ldw, c0, rows, db);
      db = nullptr;   // done inside
    }
    int rc = check_launch("fastegnn_wide_linear_dw");
    if (rc) return rc;
  }
  if (db) {
    const int gx = cdiv(O, 256), ns = row_splits(M, gx);
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, ns), dim3(256), 0, st, G, O, (long)M, O, db, (long)cdiv(M, ns));
    return check_launch("fastegnn_wide_linear_dw(bias)");
  }
  return FASTEGNN_OK;
}

// y = act(z) / dz = dy * act'(z); kind = FASTEGNN_ACT_*, p = its parameter (act_fn of the reference constructor)
int fastegnn_wide_act(const float *z, int64_t n, int32_t kind, float p, float *y, void *stream) {
  FE_REQUIRE(n >= 0 && kind >= 0 && kind <= FASTEGNN_ACT_SOFTPLUS, "fastegnn_wide_act: bad arguments");
  if (n == 0) return FASTEGNN_OK;
  FE_REQUIRE(z && y, "fastegnn_wide_act: null pointer");
  hipLaunchKernelGGL(act_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, z, (size_t)n, Act{kind, p}, y);
  return check_launch("fastegnn_wide_act");
}
int fastegnn_wide_act_backward(const float *z, const float *dy, int64_t n, int32_t kind, float p, float *dz, void *stream) {
  FE_REQUIRE(n >= 0 && kind >= 0 && kind <= FASTEGNN_ACT_SOFTPLUS, "fastegnn_wide_act_backward: bad arguments");
  if (n == 0) return FASTEGNN_OK;
  FE_REQUIRE(z && dy && dz, "fastegnn_wide_act_backward: null pointer");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream, z, dy, (size_t)n, Act{kind, p}, dz);
  return check_launch("fastegnn_wide_act_backward");
}

// out[m, :] = (base ? base[m, :] : 0) + X[idx[m], :]      (node_feat[row], virtual_node_feat[data_batch], ...)
int fastegnn_wide_gather_add(const float *X, const int64_t *idx, int64_t M, int32_t W, const float *base, float *out, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1, "fastegnn_wide_gather_add: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(X && idx && out, "fastegnn_wide_gather_add: null pointer");
  hipLaunchKernelGGL(gather_add_kernel, dim3(grid1d((size_t)M * W)), dim3(256), 0, (hipStream_t)stream, X, idx, (long)M, W, base, out);
  return check_launch("fastegnn_wide_gather_add");
}
// table[idx[m], :] += rows[m, :]                          (unsorted_segment_sum, global_mean_pool's sums; fp32 atomics)
int fastegnn_wide_scatter_add(float *table, const int64_t *idx, int64_t M, int32_t W, const float *rows, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1, "fastegnn_wide_scatter_add: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(table && idx && rows, "fastegnn_wide_scatter_add: null pointer");
  if (W >= 32) {
    const int gx = cdiv(W, 256);
    long rows_per_wg = 64;   // enough workgroups to fill the chip, runs long enough to pay
    while (rows_per_wg < 1024 && (long)gx * cdiv(M, rows_per_wg) > 16384) rows_per_wg *= 2;
    while (cdiv(M, rows_per_wg) > 65535) rows_per_wg *= 2;   // grid.y is a 16-bit quantity: beyond 67 M rows the ranges grow instead
    hipLaunchKernelGGL(scatter_add_runs_kernel, dim3(gx, (unsigned)cdiv(M, rows_per_wg)), dim3(256), 0, (hipStream_t)stream, table, idx,
                       (long)M, W, rows, rows_per_wg);
  } else {
    hipLaunchKernelGGL(scatter_add_kernel, dim3(grid1d((size_t)M * W)), dim3(256), 0, (hipStream_t)stream, table, idx, (long)M, W, rows);
  }
  return check_launch("fastegnn_wide_scatter_add");
}
// Y[m, :] = X[m, :] * s[m]                                (attention gates, 1 / count of the segment means)
int fastegnn_wide_rowscale(const float *X, const float *s, int64_t M, int32_t W, float *Y, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1, "fastegnn_wide_rowscale: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(X && s && Y, "fastegnn_wide_rowscale: null pointer");
  hipLaunchKernelGGL(rowscale_kernel, dim3(grid1d((size_t)M * W)), dim3(256), 0, (hipStream_t)stream, X, s, (long)M, W, Y);
  return check_launch("fastegnn_wide_rowscale");
}
// out[m] = <A[m, :], B[m, :]>                             (the gate's gradient)
int fastegnn_wide_rowdot(const float *A, const float *B, int64_t M, int32_t W, float *out, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1, "fastegnn_wide_rowdot: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(A && B && out, "fastegnn_wide_rowdot: null pointer");
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)cdiv(M, 16)), dim3(256), 0, (hipStream_t)stream, A, B, (long)M, W, out);
  return check_launch("fastegnn_wide_rowdot");
}

}  // extern "C"
