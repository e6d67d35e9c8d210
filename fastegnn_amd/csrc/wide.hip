// Generic-width primitives of the WIDE path: FastEGNN with 64 < hidden_nf <= 256 (reference: any --dim_hidden,
// main_nbody.py:27, models/FastEGNN.py:28-99).  The fused stage kernels of this library are built on 64-wide register
// tiles; a wider model runs UNFUSED on the operators below -- the op sequence of models/FastEGNN.py:102-223 with every
// hidden-sized tensor op as one of these launches (fastegnn_amd/wide.py assembles them, autograd composes the backward from
// the *_dx / *_dw / *_bwd entry points).  The two GEMM kernels (wide_gemm.h) run fp32-grade products as bf16x3 splits on the
// matrix pipe, with the activation of a Linear's input and the activation backward of its input gradient fused in; vector
// FMAs elsewhere, fp32 atomics for the row-keyed sums.  DESIGN.md section 9 prices this path; the tuned path is hidden_nf <= 64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#ifndef FE_ACT_GENERIC
#define FE_ACT_GENERIC   // this translation unit always carries every activation kind (act_both)
#endif
#include "kernels.h"
#include "wide_gemm.h"

namespace fe {
namespace wide {

// C[m, n] = epi(sum_k pro(A[m, k]) * B(k, n) + bias[n]) + base[m, n],  B(k, n) = Bp[k*sbk + n*sbn]   (forward of a Linear:
// B(k, o) = W[o*ldw + c0 + k]; its input gradient: B(o, k) = W[o*ldw + c0 + k]; epi = * act'(Z[m, n])) for N <= 8 columns (the
// [1, H] heads and their input gradients, the rank-1 radial columns): 32 lanes per row, each taking every 32nd group of four
// consecutive k (one 512-byte line of a 128-wide row per instruction), combined with a butterfly
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm_smalln_kernel(const float *A, int lda, long M, int Kd, const float *Bp, long sbk, long sbn,
                                                          int N, const float *bias, const float *base, float *C, int ldc,
                                                          int accumulate, Act pro, const float *Z, int ldz, Act epi) {
  constexpr int RU = 4;   // rows per 32-lane group, their loads in flight together (one row at a time ran at 2 TB/s)
  const long mg = ((long)blockIdx.x * 8 + (threadIdx.x >> 5)) * RU;
  const int sub = threadIdx.x & 31;
  const bool vec = (lda & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0;
  float s[RU][8];
#pragma unroll
  for (int u = 0; u < RU; ++u)
#pragma unroll
    for (int n = 0; n < 8; ++n) s[u][n] = 0.f;
  const int k4n = vec ? Kd >> 2 : 0;
  for (int k4 = sub; k4 < k4n; k4 += 32) {
    float4 v[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long m = mg + u < M ? mg + u : M - 1;
      v[u] = *reinterpret_cast<const float4 *>(A + (size_t)m * lda + 4 * k4);
    }
    float b[4][8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j][n] = n < N ? Bp[(size_t)(4 * k4 + j) * sbk + (size_t)n * sbn] : 0.f;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const float x[4] = {pro_t<PRO>(v[u].x, pro), pro_t<PRO>(v[u].y, pro), pro_t<PRO>(v[u].z, pro), pro_t<PRO>(v[u].w, pro)};
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (n < N) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s[u][n] += x[j] * b[j][n];
        }
    }
  }
  for (int k = 4 * k4n + sub; k < Kd; k += 32) {
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long m = mg + u < M ? mg + u : M - 1;
      const float x = pro_t<PRO>(A[(size_t)m * lda + k], pro);
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (n < N) s[u][n] += x * Bp[(size_t)k * sbk + (size_t)n * sbn];
    }
  }
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const long m = mg + u;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      if (n >= N) break;
      float t = s[u][n];
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) t += __shfl_xor(t, d, 32);
      if (m < M && sub == 0) {
        float v = t;
        if (bias) v += bias[n];
        if constexpr (EPI != AM_NONE) v *= dact_t<EPI>(Z[(size_t)m * ldz + n], epi);
        if (base) v += base[(size_t)m * ldc + n];
        float *dst = C + (size_t)m * ldc + n;
        *dst = accumulate ? *dst + v : v;
      }
    }
  }
}
// the same for a contraction of Kd <= 8 terms (the first layer's few feature columns, the input gradient of a [1, H] head): thread =
// four consecutive columns of a row when the shapes allow (N, ldc, ldz multiples of 4), bound by the write of C
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm_smallk_kernel(const float *A, int lda, long M, int Kd, const float *Bp, long sbk, long sbn,
                                                          int N, const float *bias, const float *base, float *C, int ldc,
                                                          int accumulate, Act pro, const float *Z, int ldz, Act epi, int vec) {
  const int nv = vec ? N >> 2 : N, wv = vec ? 4 : 1;
  const size_t total = (size_t)M * nv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / nv;
    const int n = (int)(i - m * nv) * wv;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < Kd; ++k) {
      const float x = pro_t<PRO>(A[m * lda + k], pro);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < wv) v[j] += x * Bp[(size_t)k * sbk + (size_t)(n + j) * sbn];
    }
    float z[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
    const float *bsrc = base ? base : (accumulate ? C : nullptr);
    if (vec) {
      if constexpr (EPI != AM_NONE) {
        const float4 t = *reinterpret_cast<const float4 *>(Z + m * ldz + n);
        z[0] = t.x; z[1] = t.y; z[2] = t.z; z[3] = t.w;
      }
      if (bsrc) {
        const float4 t = *reinterpret_cast<const float4 *>(bsrc + m * ldc + n);
        bs[0] = t.x; bs[1] = t.y; bs[2] = t.z; bs[3] = t.w;
      }
    } else {
      if constexpr (EPI != AM_NONE) z[0] = Z[m * ldz + n];
      if (bsrc) bs[0] = bsrc[m * ldc + n];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < wv) {
        if (bias) v[j] += bias[n + j];
        if constexpr (EPI != AM_NONE) v[j] *= dact_t<EPI>(z[j], epi);
        v[j] += bs[j];
      }
    if (vec) *reinterpret_cast<float4 *>(C + m * ldc + n) = float4{v[0], v[1], v[2], v[3]};
    else C[m * ldc + n] = v[0];
  }
}

// out[s*so + l*sl] += sum_m proS(S[m*lds + s]) * proL(L[m*ldl + l])   with a SMALL side (NS <= 8 columns): the [1, H] heads' weight
// gradients (S = G, L = X) and the few feature columns of a first layer (S = X, L = G); the activation applies to the side that is
// X.  Thread = VW (4 or 1) consecutive columns l in one of the workgroup's row slots; a workgroup walks its row range with every
// slot, the slots' partials (double: cancelling column sums over up to 10^6 rows) meet in LDS, one fp32 atomic per column and range.
template <int NS, int VW, int PS, int PL>
__global__ __launch_bounds__(256) void tn_small_kernel(const float *S, int lds_, int ns, const float *L, int ldl, int nl, long M,
                                                       float *out, long so, long sl, long rows_per_split, int cols_per_wg, Act proS,
                                                       Act proL) {
  __shared__ double red[256 * VW];
  const long r_lo = (long)blockIdx.y * rows_per_split, r_hi = r_lo + rows_per_split < M ? r_lo + rows_per_split : M;
  const int slots = 256 / cols_per_wg, slot = threadIdx.x / cols_per_wg, cw = threadIdx.x % cols_per_wg;
  const int l = (blockIdx.x * cols_per_wg + cw) * VW;
  const bool live = l < nl;
  double acc[NS][VW];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int j = 0; j < VW; ++j) acc[s][j] = 0.0;
  if (live) {
#pragma unroll 4
    for (long m = r_lo + slot; m < r_hi; m += slots) {
      float x[VW];
      if constexpr (VW == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(L + (size_t)m * ldl + l);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
      } else x[0] = L[(size_t)m * ldl + l];
#pragma unroll
      for (int j = 0; j < VW; ++j) x[j] = pro_t<PL>(x[j], proL);
#pragma unroll
      for (int s = 0; s < NS; ++s)
        if (s < ns) {
          const double sv = (double)pro_t<PS>(S[(size_t)m * lds_ + s], proS);
#pragma unroll
          for (int j = 0; j < VW; ++j) acc[s][j] += sv * (double)x[j];
        }
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s >= ns) break;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VW; ++j) red[(slot * cols_per_wg + cw) * VW + j] = acc[s][j];
    __syncthreads();
    if (slot == 0 && live) {
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        double t = 0.0;
        for (int q = 0; q < slots; ++q) t += red[(q * cols_per_wg + cw) * VW + j];
        atomicAdd(out + (size_t)s * so + (size_t)(l + j) * sl, (float)t);
      }
    }
  }
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

// out[m, :] = (base ? base[m, :] : 0) + P[i1[m], :] + (Q ? Q[i2[m], :] : 0) + sum_{k < nf} feat[m, k] * Wf[w*ldw + k]
// The first Linear of edge_model / edge_mode_virtual over its torch.cat input (models/FastEGNN.py:102-119) after the node-sized
// products: P[row] + Q[col] + radial / edge_attr columns in ONE pass that only writes [M, W] (three launches and two
// read-modify-write passes before).  Thread = VW (4 or 1) consecutive columns, the workgroup's threads tile rows x column groups.
// EPI: the sum times act'(Z[m, :]) -- the backward of y = act(z) whose consumers were y itself and a segment sum of y
// (fastegnn_wide_act_scatter): dz = (g_y + g_sum[idx]) * act'(z) in one pass.
template <int VW, int EPI>
__global__ __launch_bounds__(256) void gather2_kernel(const float *P, const int64_t *i1, const float *Q, const int64_t *i2, const float *feat,
                                                      int nf, const float *Wf, int ldw, const float *base, float *out, long M, int W,
                                                      int cols_per_wg, const float *Z, Act epi) {
  const int slots = 256 / cols_per_wg, slot = threadIdx.x / cols_per_wg, cw = threadIdx.x % cols_per_wg;
  const int w = (blockIdx.y * cols_per_wg + cw) * VW;   // (rows wider than a workgroup: column blocks on grid.y)
  if (w >= W) return;
  float wf[8][VW];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < VW; ++j) wf[k][j] = (feat && k < nf) ? Wf[(size_t)(w + j) * ldw + k] : 0.f;
  auto ld = [&](const float *p, float (&v)[VW]) {
    if constexpr (VW == 4) {
      const float4 t = *reinterpret_cast<const float4 *>(p);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else v[0] = *p;
  };
  const long stride = (long)gridDim.x * slots;
#pragma unroll 2
  for (long m = (long)blockIdx.x * slots + slot; m < M; m += stride) {
    float a[VW], b[VW], c[VW];
    ld(P + (size_t)i1[m] * W + w, a);
    if (Q) ld(Q + (size_t)i2[m] * W + w, b);
    if (base) ld(base + (size_t)m * W + w, c);
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      if (Q) a[j] += b[j];
      if (base) a[j] += c[j];
    }
    if (feat) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < nf) {
          const float f = feat[(size_t)m * nf + k];
#pragma unroll
          for (int j = 0; j < VW; ++j) a[j] += f * wf[k][j];
        }
    }
    if constexpr (EPI != AM_NONE) {
      float z[VW];
      ld(Z + (size_t)m * W + w, z);
#pragma unroll
      for (int j = 0; j < VW; ++j) a[j] *= dact_t<EPI>(z[j], epi);
    }
    if constexpr (VW == 4) *reinterpret_cast<float4 *>(out + (size_t)m * W + w) = float4{a[0], a[1], a[2], a[3]};
    else out[(size_t)m * W + w] = a[0];
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
// table[idx[m], :] += rows[perm ? perm[m] : m, :] for wide rows: thread = VW consecutive columns in one of the workgroup's row slots;
// a slot walks its contiguous share of the workgroup's row range in order with a running sum that is flushed (one atomic per column)
// whenever the target changes -- sorted or grouped indices (segment sums by graph, by edge row, the C channel rows of a node) cost
// one atomic per run instead of one per row.  An UNSORTED index (the edge columns) gets the same treatment through `perm`, the
// permutation that sorts it, computed once per graph: idx then holds the sorted targets and the rows are read through perm.
// PRO: rows are PRE-activations -- y = act(rows) is what gets summed, and it is stored to `yout` on the way (the activation and
// the segment sum of its output in one pass: fastegnn_wide_act_scatter).
template <int VW, int PRO>
__global__ __launch_bounds__(256) void scatter_add_runs_kernel(float *table, const int64_t *idx, const int64_t *perm, long M, int W,
                                                               const float *rows, long rows_per_slot, int cols_per_wg, Act pro,
                                                               float *yout) {
  const int slots = 256 / cols_per_wg, slot = threadIdx.x / cols_per_wg, cw = threadIdx.x % cols_per_wg;
  const int w = (blockIdx.y * cols_per_wg + cw) * VW;
  const long m_lo = ((long)blockIdx.x * slots + slot) * rows_per_slot, m_hi = m_lo + rows_per_slot < M ? m_lo + rows_per_slot : M;
  // the slots' LAST runs meet in LDS: consecutive slots that end on the same target (sums by graph: every slot of the workgroup)
  // cost one atomic per column together -- with one graph in the batch all N rows land on one row of the table, and an atomic
  // per slot and column was 3 000 serialised atomics per address
  __shared__ float tail[256 * VW];
  __shared__ long long tcur[256];
  const bool live = w < W && m_lo < m_hi;
  int64_t cur = live ? idx[m_lo] : -1;
  float acc[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) acc[j] = 0.f;
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < VW; ++j) atomicAdd(table + (size_t)cur * W + w + j, acc[j]);
  };
  for (long m = m_lo; live && m < m_hi; m += 4) {
    int64_t t[4];
    float v[4][VW];
    // the four rows' loads first: the atomics of a flush order every later load behind them
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long mm = m + u < m_hi ? m + u : m_hi - 1;
      t[u] = idx[mm];
      const size_t off = (size_t)(perm ? perm[mm] : mm) * W + w;
      const float *src = rows + off;
      if constexpr (VW == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(src);
        v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
      } else v[u][0] = *src;
      if constexpr (PRO != AM_NONE) {
#pragma unroll
        for (int j = 0; j < VW; ++j) v[u][j] = pro_t<PRO>(v[u][j], pro);
        if (m + u < m_hi) {
          if constexpr (VW == 4) *reinterpret_cast<float4 *>(yout + off) = float4{v[u][0], v[u][1], v[u][2], v[u][3]};
          else yout[off] = v[u][0];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (m + u >= m_hi) break;
      if (t[u] != cur) {
        flush();
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] = 0.f;
        cur = t[u];
      }
#pragma unroll
      for (int j = 0; j < VW; ++j) acc[j] += v[u][j];
    }
  }
#pragma unroll
  for (int j = 0; j < VW; ++j) tail[threadIdx.x * VW + j] = acc[j];
  if (cw == 0) tcur[slot] = w < W ? (long long)cur : -1;
  __syncthreads();
  if (live && (slot == 0 || tcur[slot - 1] != cur)) {   // the first slot of a group of equal last targets adds the group's tails
    for (int s2 = slot + 1; s2 < slots && tcur[s2] == cur; ++s2) {
#pragma unroll
      for (int j = 0; j < VW; ++j) acc[j] += tail[(s2 * cols_per_wg + cw) * VW + j];
    }
    flush();
  }
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
// row ranges of the reductions over M: enough workgroups to fill the chip, at least `min_rows` rows each
inline int row_splits(long M, long other_wgs, long min_rows = 512, long target = 2048) {
  long want = (target + other_wgs - 1) / (other_wgs > 0 ? other_wgs : 1);
  long most = (M + min_rows - 1) / min_rows;
  if (want > most) want = most;
  if (want < 1) want = 1;
  if (want > 16384) want = 16384;
  return (int)want;
}
inline int pow2_at_least(int n) {
  int p = 1;
  while (p < n) p *= 2;
  return p;
}
inline bool act_ok(int kind) { return kind >= ACT_NONE && kind <= FASTEGNN_ACT_SOFTPLUS; }

// can the head's output ride in the epilogue of its first Linear?  One column block of the streaming GEMM: 9 .. 128 hidden units,
// an input the 16-byte loads can take
static bool head_forward_fits(int lda, int Kd, int N, const float *A) {
  return N > 8 && N <= 128 && Kd > 8 && !(lda & 3) && !(Kd & 3) && !(reinterpret_cast<size_t>(A) & 15);
}
static int gemm(const float *A, int lda, long M, int Kd, const float *Bp, long sbk, long sbn, int N, const float *bias,
                const float *base, float *C, int ldc, int accumulate, Act pro, const float *Z, int ldz, Act epi, hipStream_t st,
                const char *what, const float *gs = nullptr, const float *w2 = nullptr, float *sout = nullptr, const float *b2 = nullptr) {
  if (M == 0 || N == 0) return FASTEGNN_OK;
  // gs: the generated-operand form of a scalar head's backward (AM_HEAD_*): A is the head's stored pre-activation
  // sout: the head's output from the GEMM's accumulators (AM_DOT_*; `epi` holds the head's activation, w2 its second weight)
  const int pm = gs ? (am_of(pro.kind) == AM_SILU ? AM_HEAD_SILU : AM_HEAD_GEN) : am_of(pro.kind);
  const int em = sout ? (am_of(epi.kind) == AM_SILU ? AM_DOT_SILU : AM_DOT_GEN) : (Z ? am_of(epi.kind) : AM_NONE);
  FE_REQUIRE(!sout || head_forward_fits(lda, Kd, N, A), "fastegnn_wide_head_forward: internal dispatch");
  FE_REQUIRE(!gs || (N > 8 && Kd > 8 && !(lda & 3) && !(Kd & 3) && !(reinterpret_cast<size_t>(A) & 15)),
             "fastegnn_wide_head_dx: widths of at least 9, the head's a multiple of 4");
  FE_REQUIRE(!(pm && em) && !(base && accumulate), "wide gemm: unsupported combination of fused steps");
// one launch per (prologue, epilogue) mode: the two never meet in one call
#define FE_MODES(LAUNCH)                                 \
  do {                                                   \
    if (pm == AM_SILU) { LAUNCH(AM_SILU, AM_NONE); }     \
    else if (pm == AM_GEN) { LAUNCH(AM_GEN, AM_NONE); }  \
    else if (em == AM_SILU) { LAUNCH(AM_NONE, AM_SILU); } \
    else if (em == AM_GEN) { LAUNCH(AM_NONE, AM_GEN); }  \
    else { LAUNCH(AM_NONE, AM_NONE); }                   \
  } while (0)
  if (N <= 8) {
#define FE_SMALLN(P_, E_)                                                                                                            \
  hipLaunchKernelGGL((gemm_smalln_kernel<P_, E_>), dim3((unsigned)cdiv(M, 32)), dim3(256), 0, st, A, lda, M, Kd, Bp, sbk, sbn, N, bias, \
                     base, C, ldc, accumulate, pro, Z, ldz, epi)
    FE_MODES(FE_SMALLN);
#undef FE_SMALLN
  } else if (Kd <= 8 || (lda & 3) || (Kd & 3) || (reinterpret_cast<size_t>(A) & 15)) {
    // few terms -- or a row stride the streaming GEMM's 16-byte loads cannot take (K not a multiple of 4: rare, any speed will do)
    const int vec = (N & 3) == 0 && (ldc & 3) == 0 && (!Z || (ldz & 3) == 0) && (reinterpret_cast<size_t>(C) & 15) == 0 &&
                    (reinterpret_cast<size_t>(base) & 15) == 0 && (reinterpret_cast<size_t>(Z) & 15) == 0;
#define FE_SMALLK(P_, E_)                                                                                                              \
  hipLaunchKernelGGL((gemm_smallk_kernel<P_, E_>), dim3(grid1d((size_t)M * (vec ? N / 4 : N))), dim3(256), 0, st, A, lda, M, Kd, Bp, sbk, \
                     sbn, N, bias, base, C, ldc, accumulate, pro, Z, ldz, epi, vec)
    FE_MODES(FE_SMALLK);
#undef FE_SMALLK
  } else {
    // The four-buffer prefetch (DEEP) exists for full 128-column blocks and panels of four chunks; it takes every shape whose
    // padding to those costs less than a third of the work (96, 128, 192 .. 256 wide; 160 = 128 + 32 columns / 5 + 3 chunks does
    // not).  Otherwise: column blocks of 32 NQ <= 128 columns, as even as the width allows (160 = 96 + 64, not 128 + 32).
    const int nquads = cdiv(N, 32), chunks = cdiv(Kd, 32);
    const bool deep = 3 * cdiv(nquads, 4) * 4 <= 4 * nquads && 3 * cdiv(chunks, 4) * 4 <= 4 * chunks;
    const int nblocks = cdiv(nquads, 4), nq = deep ? 4 : cdiv(nquads, nblocks), gy = cdiv(nquads, nq);
    const long units = cdiv(M, 32);
    const int wgs_per_cu = (x3_lds_bytes(nq) + (gs ? 4 * Kd + 128 : 0)) * 2 <= 160 * 1024 ? 2 : 1;
    long gx = cdiv(units, XWAVES);
    if (gx > 256 * wgs_per_cu) gx = 256 * wgs_per_cu;
    GemmX3 g{A, lda, M, Kd, Bp, sbk, sbn, N, bias, base, C, ldc, accumulate, pro, Z, ldz, epi, (int)cdiv(units, gx * XWAVES), gs, w2, sout, b2};
    const dim3 grid((unsigned)gx, (unsigned)gy);
    FE_REQUIRE(launch_gemm_x3(g, nq, pm, em, deep, grid, st) == 0,
               "fastegnn_wide_head_dx: hidden and input widths of 96, 128, 192 .. 256 (+ multiples of 128) only");
  }
#undef FE_MODES
  return check_launch(what);
}

}  // namespace wide
}  // namespace fe

using namespace fe;
using namespace fe::wide;

extern "C" {

// out[M, O] = (base ? base : 0) + act(X)[M, K] . W[:, c0 : c0 + K]^T + bias        (models/FastEGNN.py: every nn.Linear; a Linear over
// a torch.cat of inputs is the sum of these calls over the weight's column blocks, chained through `base`).  act_kind >= 0: X is
// the PRE-activation of the Linear's input (nn.Sequential(Linear, act, Linear): the activation runs in this kernel's prologue and
// its output never reaches memory); FASTEGNN_ACT_NONE: X as is.
int fastegnn_wide_linear(const float *X, int64_t M, int32_t K, const float *W, int32_t ldw, int32_t c0, const float *bias,
                         const float *base, float *out, int32_t O, int32_t act_kind, float act_p, void *stream) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0 && act_ok(act_kind), "fastegnn_wide_linear: bad arguments");
  FE_REQUIRE((X || M == 0) && W && (out || M == 0), "fastegnn_wide_linear: null pointer");
  return gemm(X, K, M, K, W + c0, 1, ldw, O, bias, base, out, O, 0, Act{act_kind, act_p}, nullptr, 0, Act{ACT_NONE, 0.f},
              (hipStream_t)stream, "fastegnn_wide_linear");
}
// dX[M, K] (+)= (G[M, O] . W[:, c0 : c0 + K]) * act'(Z[M, K])      (Z may be null: no activation; with Z the result is the
// gradient of the PRE-activation Z that fastegnn_wide_linear took with the same act_kind)
int fastegnn_wide_linear_dx(const float *G, int64_t M, int32_t O, const float *W, int32_t ldw, int32_t c0, int32_t K, float *dX,
                            int32_t accumulate, const float *Z, int32_t act_kind, float act_p, void *stream) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0 && act_ok(act_kind), "fastegnn_wide_linear_dx: bad arguments");
  FE_REQUIRE((G || M == 0) && W && (dX || M == 0), "fastegnn_wide_linear_dx: null pointer");
  FE_REQUIRE(!Z || act_kind >= 0, "fastegnn_wide_linear_dx: Z needs an activation kind");
  return gemm(G, O, M, O, W + c0, ldw, 1, K, nullptr, nullptr, dX, K, accumulate, Act{ACT_NONE, 0.f}, Z, K, Act{act_kind, act_p},
              (hipStream_t)stream, "fastegnn_wide_linear_dx");
}
// dW[:, c0 : c0 + K] += G[M, O]^T . act(X)[M, K];  db[O] += column sums of G (db may be null)
static int linear_dw(const float *G, const float *X, int64_t M, int32_t O, int32_t K, float *dW, int32_t ldw, int32_t c0, float *db,
                     int32_t act_kind, float act_p, void *stream, const float *gs, const float *w2, Act gen, float *dw2 = nullptr) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0 && act_ok(act_kind), "fastegnn_wide_linear_dw: bad arguments");
  FE_REQUIRE((G && X) || M == 0, "fastegnn_wide_linear_dw: null pointer");
  FE_REQUIRE(!gs || (dW && O > 8 && K > 8), "fastegnn_wide_head_dw: widths of at least 9");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) return FASTEGNN_OK;
  const Act pro{act_kind, act_p};
  if (dW) {
    if (O <= 8 || K <= 8) {
      // small side S = G's columns and long side L = X's, or the other way round; the activation belongs to the X side
      const bool g_small = O <= 8;
      const float *S = g_small ? G : X, *Lg = g_small ? X : G;
      const int nsm = g_small ? O : K, nl = g_small ? K : O;
      const long so = g_small ? ldw : 1, sl = g_small ? 1 : ldw;
      const int vw = ((nl & 3) == 0 && (reinterpret_cast<size_t>(Lg) & 15) == 0) ? 4 : 1;
      const int nlv = nl / vw;
      const int cols = nlv >= 256 ? 256 : pow2_at_least(nlv);
      const int gx = cdiv(nlv, cols), nsplit = row_splits(M, gx, 256 / cols * 64, 1024);
      const dim3 grid(gx, nsplit);
      const long rps = cdiv(M, nsplit);
      const int am = am_of(act_kind), ps = g_small ? AM_NONE : am, pl = g_small ? am : AM_NONE;
#define FE_TNS(NS_, VW_, PS_, PL_)                                                                                                    \
  hipLaunchKernelGGL((tn_small_kernel<NS_, VW_, PS_, PL_>), grid, dim3(256), 0, st, S, nsm, nsm, Lg, nl, nl, (long)M, dW + c0, so, sl, rps, \
                     cols, pro, pro)
#define FE_TNS_ACT(NS_, VW_)                                           \
  do {                                                                 \
    if (ps == AM_SILU) FE_TNS(NS_, VW_, AM_SILU, AM_NONE);             \
    else if (ps == AM_GEN) FE_TNS(NS_, VW_, AM_GEN, AM_NONE);          \
    else if (pl == AM_SILU) FE_TNS(NS_, VW_, AM_NONE, AM_SILU);        \
    else if (pl == AM_GEN) FE_TNS(NS_, VW_, AM_NONE, AM_GEN);          \
    else FE_TNS(NS_, VW_, AM_NONE, AM_NONE);                           \
  } while (0)
      if (vw == 4) {
        if (nsm == 1) FE_TNS_ACT(1, 4);
        else if (nsm <= 4) FE_TNS_ACT(4, 4);
        else FE_TNS_ACT(8, 4);
      } else {
        if (nsm == 1) FE_TNS_ACT(1, 1);
        else FE_TNS_ACT(8, 1);
      }
#undef FE_TNS_ACT
#undef FE_TNS
    } else {
      const int gx = cdiv(O, TB), gy = cdiv(K, TB), ns = row_splits(M, (long)gx * gy, 256, 512);
      long rows = cdiv(M, ns);
      rows = (rows + 31) / 32 * 32;
      TnX3 t{G, O, X, K, (long)M, O, K, dW, ldw, c0, rows, db, pro, gs, w2, gen, dw2};
      const dim3 grid(gx, gy, (unsigned)cdiv(M, rows));
      launch_tn_x3(t, am_of(act_kind), gs ? am_of(gen.kind) : AM_NONE, grid, st);
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
int fastegnn_wide_linear_dw(const float *G, const float *X, int64_t M, int32_t O, int32_t K, float *dW, int32_t ldw, int32_t c0,
                            float *db, int32_t act_kind, float act_p, void *stream) {
  return linear_dw(G, X, M, O, K, dW, ldw, c0, db, act_kind, act_p, stream, nullptr, nullptr, Act{ACT_NONE, 0.f});
}
// The backward of the FIRST Linear of a scalar head  s = act(X W1^T + b1) . w2^T (+ b2)  (coord_mlp_r / _r_virtual / _v_virtual / _vel,
// gravity_mlp: models/FastEGNN.py:55-99) straight from the head's output gradient gs [M]: the gradient of the hidden
// pre-activation, G[m, o] = gs[m] * w2[o] * act'(Zc[m, o]), is formed inside the kernels from the stored Zc and never written.
//   head_dx   dX[M, K] (+)= G . W1[:, c0 : c0 + K]
//   head_dw   dW1[:, c0 : c0 + K] += G^T act_x(X),  db1 += column sums of G     (x_kind: FASTEGNN_ACT_NONE or X a pre-activation),
//             dw2[o] += sum_m gs[m] act(Zc[m, o]) (the second Linear's weight gradient, from the same pass; may be NULL)
// O (the head's hidden width) and K at least 9, O a multiple of 4.
//   head_forward   Zc = act_x(X) . W1[:, c0 : c0 + K]^T + b1 stored, s[m] = act(Zc[m, :]) . w2 + (b2 ? b2[0] : 0): with 9 .. 128 hidden
//             units the output is formed from the first GEMM's accumulators, otherwise by a second launch over Zc
int fastegnn_wide_head_dx(const float *gs, const float *w2, const float *Zc, int64_t M, int32_t O, const float *W, int32_t ldw,
                          int32_t c0, int32_t K, float *dX, int32_t accumulate, int32_t kind, float p, void *stream) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0 && kind >= 0 && kind <= FASTEGNN_ACT_SOFTPLUS,
             "fastegnn_wide_head_dx: bad arguments");
  FE_REQUIRE(M == 0 || (gs && w2 && Zc && W && dX), "fastegnn_wide_head_dx: null pointer");
  return gemm(Zc, O, M, O, W + c0, ldw, 1, K, nullptr, nullptr, dX, K, accumulate, Act{kind, p}, nullptr, 0, Act{ACT_NONE, 0.f},
              (hipStream_t)stream, "fastegnn_wide_head_dx", gs, w2);
}
int fastegnn_wide_head_dw(const float *gs, const float *w2, const float *Zc, const float *X, int64_t M, int32_t O, int32_t K, float *dW,
                          int32_t ldw, int32_t c0, float *db, float *dw2, int32_t kind, float p, int32_t x_kind, float x_p, void *stream) {
  FE_REQUIRE(kind >= 0 && kind <= FASTEGNN_ACT_SOFTPLUS && (M == 0 || (gs && w2)), "fastegnn_wide_head_dw: bad arguments");
  return linear_dw(Zc, X, M, O, K, dW, ldw, c0, db, x_kind, x_p, stream, gs, w2, Act{kind, p}, dw2);
}
int fastegnn_wide_head_forward(const float *X, int64_t M, int32_t K, const float *W1, int32_t ldw, int32_t c0, const float *b1,
                               const float *w2, const float *b2, float *Zc, float *s, int32_t O, int32_t kind, float p, int32_t x_kind,
                               float x_p, void *stream) {
  FE_REQUIRE(M >= 0 && K >= 1 && O >= 1 && ldw >= c0 + K && c0 >= 0 && kind >= 0 && kind <= FASTEGNN_ACT_SOFTPLUS && act_ok(x_kind),
             "fastegnn_wide_head_forward: bad arguments");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(X && W1 && w2 && Zc && s, "fastegnn_wide_head_forward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const Act act{kind, p}, xact{x_kind, x_p}, none{ACT_NONE, 0.f};
  if (head_forward_fits(K, K, O, X))
    return gemm(X, K, M, K, W1 + c0, 1, ldw, O, b1, nullptr, Zc, O, 0, xact, nullptr, 0, act, st, "fastegnn_wide_head_forward", nullptr, w2, s, b2);
  int rc = gemm(X, K, M, K, W1 + c0, 1, ldw, O, b1, nullptr, Zc, O, 0, xact, nullptr, 0, none, st, "fastegnn_wide_head_forward");
  if (rc) return rc;
  return gemm(Zc, O, M, O, w2, 1, O, 1, b2, nullptr, s, 1, 0, act, nullptr, 0, none, st, "fastegnn_wide_head_forward(out)");
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

static int gather2(const float *P, const int64_t *i1, const float *Q, const int64_t *i2, const float *feat, int nf, const float *Wf,
                   int ldw, const float *base, float *out, long M, int W, hipStream_t st, const char *what, const float *Z = nullptr,
                   Act epi = Act{ACT_NONE, 0.f}) {
  const bool v4 = (W & 3) == 0 && ((reinterpret_cast<size_t>(P) | reinterpret_cast<size_t>(Q) | reinterpret_cast<size_t>(base) |
                                    reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(Z)) & 15) == 0;
  const int nv = v4 ? W / 4 : W;
  const int cols = nv >= 256 ? 256 : pow2_at_least(nv), slots = 256 / cols, gy = cdiv(nv, cols);
  long gx = cdiv(M, (long)slots * 4);
  if (gx > 8192 / gy) gx = 8192 / gy;
  if (gx < 1) gx = 1;
  const int em = Z ? am_of(epi.kind) : AM_NONE;
#define FE_G2(VW_, E_) \
  hipLaunchKernelGGL((gather2_kernel<VW_, E_>), dim3((unsigned)gx, gy), dim3(256), 0, st, P, i1, Q, i2, feat, nf, Wf, ldw, base, out, M, W, cols, Z, epi)
  if (v4) {
    if (em == AM_SILU) FE_G2(4, AM_SILU);
    else if (em == AM_GEN) FE_G2(4, AM_GEN);
    else FE_G2(4, AM_NONE);
  } else {
    if (em == AM_SILU) FE_G2(1, AM_SILU);
    else if (em == AM_GEN) FE_G2(1, AM_GEN);
    else FE_G2(1, AM_NONE);
  }
#undef FE_G2
  return check_launch(what);
}
// (narrow rows too: the 3-vector sums by graph put N rows on a handful of addresses -- as runs they cost a few atomics per slot;
//  FASTEGNN_WIDE_SCATTER_ATOMICS=1 brings the one-atomic-per-element kernel back for narrow rows: A/B lever)
static int scatter_add(float *table, const int64_t *idx, const int64_t *perm, long M, int W, const float *rows, hipStream_t st,
                       const char *what, Act pro = Act{ACT_NONE, 0.f}, float *yout = nullptr) {
  static const bool atomics_only = getenv("FASTEGNN_WIDE_SCATTER_ATOMICS") && atoi(getenv("FASTEGNN_WIDE_SCATTER_ATOMICS")) != 0;
  const bool v4 = (W & 3) == 0 && ((reinterpret_cast<size_t>(rows) | reinterpret_cast<size_t>(table) | reinterpret_cast<size_t>(yout)) & 15) == 0;
  const int nv = v4 ? W / 4 : W;
  const int pm = yout ? am_of(pro.kind) : AM_NONE;
  if (W >= 32 || !atomics_only || pm) {
    const int cols = nv >= 256 ? 256 : pow2_at_least(nv), slots = 256 / cols, gy = cdiv(nv, cols);
    long rps = 16;   // rows per slot: enough workgroups to fill the chip, runs long enough to pay (and few atomics per address)
    while (rps < 256 && cdiv(M, rps * slots) * gy > 2048) rps *= 2;
    const long gx = cdiv(M, rps * slots);
    FE_REQUIRE(gx <= 0x7fffffffL && gy <= 65535, "fastegnn_wide_scatter_add: too many rows or columns");
#define FE_SC(VW_, P_) \
  hipLaunchKernelGGL((scatter_add_runs_kernel<VW_, P_>), dim3((unsigned)gx, gy), dim3(256), 0, st, table, idx, perm, M, W, rows, rps, cols, pro, yout)
    if (v4) {
      if (pm == AM_SILU) FE_SC(4, AM_SILU);
      else if (pm == AM_GEN) FE_SC(4, AM_GEN);
      else FE_SC(4, AM_NONE);
    } else {
      if (pm == AM_SILU) FE_SC(1, AM_SILU);
      else if (pm == AM_GEN) FE_SC(1, AM_GEN);
      else FE_SC(1, AM_NONE);
    }
#undef FE_SC
  } else {
    FE_REQUIRE(!perm, "fastegnn_wide_scatter_add_perm: not with FASTEGNN_WIDE_SCATTER_ATOMICS");
    hipLaunchKernelGGL(scatter_add_kernel, dim3(grid1d((size_t)M * W)), dim3(256), 0, st, table, idx, M, W, rows);
  }
  return check_launch(what);
}

// out[m, :] = (base ? base[m, :] : 0) + X[idx[m], :]      (node_feat[row], virtual_node_feat[data_batch], ...)
int fastegnn_wide_gather_add(const float *X, const int64_t *idx, int64_t M, int32_t W, const float *base, float *out, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1, "fastegnn_wide_gather_add: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(X && idx && out, "fastegnn_wide_gather_add: null pointer");
  return gather2(X, idx, nullptr, nullptr, nullptr, 0, nullptr, 0, base, out, M, W, (hipStream_t)stream, "fastegnn_wide_gather_add");
}
// out[m, :] = (base ? base[m, :] : 0) + P[i1[m], :] + (Q ? Q[i2[m], :] : 0) + feat[m, 0:nf] . Wf[:, c0 : c0 + nf]^T   (feat may be
// NULL; nf <= 8): the first Linear of edge_model over cat[h[row], h[col], radial, edge_attr] (models/FastEGNN.py:102-108) given
// P = h . W[:, 0:H]^T + b and Q = h . W[:, H:2H]^T, and of edge_mode_virtual (:111-119) in the same form
int fastegnn_wide_gather2(const float *P, const int64_t *i1, const float *Q, const int64_t *i2, const float *feat, int32_t nf,
                          const float *Wf, int32_t ldw, int32_t c0, const float *base, float *out, int64_t M, int32_t W, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1 && nf >= 0 && nf <= 8, "fastegnn_wide_gather2: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(P && i1 && out && (!Q || i2) && (!feat || nf == 0 || (Wf && ldw >= c0 + nf && c0 >= 0)), "fastegnn_wide_gather2: bad pointers");
  return gather2(P, i1, Q, i2, nf ? feat : nullptr, nf, Wf ? Wf + c0 : nullptr, ldw, base, out, M, W, (hipStream_t)stream,
                 "fastegnn_wide_gather2");
}
// table[idx[m], :] += rows[m, :]                          (unsorted_segment_sum, global_mean_pool's sums; fp32 atomics)
int fastegnn_wide_scatter_add(float *table, const int64_t *idx, int64_t M, int32_t W, const float *rows, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1, "fastegnn_wide_scatter_add: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(table && idx && rows, "fastegnn_wide_scatter_add: null pointer");
  return scatter_add(table, idx, nullptr, M, W, rows, (hipStream_t)stream, "fastegnn_wide_scatter_add");
}
// y = act(z) stored, and table[idx[m], :] += y[m, :] in the same pass (the activation of edge_mlp / edge_mlp_virtual's output and
// the segment sum of it that node_model / node_model_virtual start with, models/FastEGNN.py:108-119, 156, 170)
int fastegnn_wide_act_scatter(const float *z, const int64_t *idx, int64_t M, int32_t W, int32_t kind, float p, float *y, float *table,
                              void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1 && kind >= 0 && kind <= FASTEGNN_ACT_SOFTPLUS, "fastegnn_wide_act_scatter: bad arguments");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(z && idx && y && table, "fastegnn_wide_act_scatter: null pointer");
  return scatter_add(table, idx, nullptr, M, W, z, (hipStream_t)stream, "fastegnn_wide_act_scatter", Act{kind, p}, y);
}
// dz[m, :] = ((g_y ? g_y[m, :] : 0) + g_table[idx[m], :]) * act'(z[m, :]): its backward
int fastegnn_wide_act_scatter_backward(const float *z, const int64_t *idx, int64_t M, int32_t W, int32_t kind, float p, const float *g_y,
                                       const float *g_table, float *dz, void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1 && kind >= 0 && kind <= FASTEGNN_ACT_SOFTPLUS, "fastegnn_wide_act_scatter_backward: bad arguments");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(z && idx && g_table && dz, "fastegnn_wide_act_scatter_backward: null pointer");
  return gather2(g_table, idx, nullptr, nullptr, nullptr, 0, nullptr, 0, g_y, dz, M, W, (hipStream_t)stream,
                 "fastegnn_wide_act_scatter_backward", z, Act{kind, p});
}
// table[idx_sorted[m], :] += rows[perm[m], :]: the same sums for an index in any order, given the permutation that sorts it
// (idx_sorted[m] = idx[perm[m]], computed once per graph) -- runs of equal targets become one atomic each
int fastegnn_wide_scatter_add_perm(float *table, const int64_t *idx_sorted, const int64_t *perm, int64_t M, int32_t W, const float *rows,
                                   void *stream) {
  FE_REQUIRE(M >= 0 && W >= 1, "fastegnn_wide_scatter_add_perm: bad sizes");
  if (M == 0) return FASTEGNN_OK;
  FE_REQUIRE(table && idx_sorted && perm && rows, "fastegnn_wide_scatter_add_perm: null pointer");
  return scatter_add(table, idx_sorted, perm, M, W, rows, (hipStream_t)stream, "fastegnn_wide_scatter_add_perm");
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
