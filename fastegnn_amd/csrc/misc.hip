// Small kernels: model prologue/epilogue, row permutation, generic weight-gradient GEMMs,
// and the toolkit self-test.
#include <cstdlib>
#include "kernels.h"

namespace fe {

// ---------------------------------------------------------------- embedding (FastEGNN.py:271)
__global__ void embed_fwd_kernel(const float *nf, int N, int nfd, const float *W, const float *b, float *h) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * H) return;
  int n = (int)(idx >> 6), o = (int)(idx & 63);
  float acc = b[o];
  for (int a = 0; a < nfd; ++a) acc += nf[(size_t)n * nfd + a] * W[o * nfd + a];
  h[idx] = acc;
}
// g_in[n, a] (+)= sum_o G[n, o] * W[o * ldw + c0 + a]: the input gradient of a narrow column block of a 64-row Linear
// (embedding_in: all nf columns; node_mlp.0: the node_attr columns).  N * kf threads, a few MB of traffic.
__global__ void dgrad_small_kernel(const float *G, long N, int kf, const float *W, int ldw, int c0, float *g_in, int accumulate) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * kf) return;
  long n = idx / kf;
  int a = (int)(idx % kf);
  float acc = 0.f;
  for (int o = 0; o < H; ++o) acc += G[(size_t)n * H + o] * W[(size_t)o * ldw + c0 + a];
  g_in[idx] = accumulate ? g_in[idx] + acc : acc;
}

// ---------------------------------------------------------------- virtual_node_feat.repeat (:268)
__global__ void virtual_init_kernel(const float *vnf, int B, int C, float *HvT) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * C * H) return;
  int hh = (int)(idx & 63);
  int c = (int)((idx >> 6) % C);
  HvT[idx] = vnf[hh * C + c];
}
__global__ void virtual_init_bwd_kernel(const float *g_HvT, int B, int C, float *g_vnf) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * H) return;
  int hh = idx & 63, c = idx >> 6;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += g_HvT[((size_t)b * C + c) * H + hh];
  g_vnf[hh * C + c] += acc;
}

// ---------------------------------------------------------------- hidden_nf < 64: zero-padded parameter images
constexpr int PAD_MAX_DESC = 64;
struct PadTable {
  fastegnn_pad_desc_t d[PAD_MAX_DESC];
  int n, h;
};
// source column of padded column c (or -1: a zero of the padding)
__device__ __forceinline__ int pad_src_col(const fastegnn_pad_desc_t &d, int h, int c) {
  if (c < d.lead) return c;
  int os = d.lead, od = d.lead;
  for (int b = 0; b < d.nblk; ++b) {
    const int wd = d.blk[b] / h * H;
    if (c < od + wd) return c - od < d.blk[b] ? os + (c - od) : -1;
    os += d.blk[b];
    od += wd;
  }
  return os + (c - od);
}
__device__ __forceinline__ int pad_dst_col(const fastegnn_pad_desc_t &d, int h, int c) {
  if (c < d.lead) return c;
  int os = d.lead, od = d.lead;
  for (int b = 0; b < d.nblk; ++b) {
    if (c < os + d.blk[b]) return od + (c - os);
    os += d.blk[b];
    od += d.blk[b] / h * H;
  }
  return od + (c - os);
}
__global__ __launch_bounds__(256) void pad_params_kernel(PadTable tab, int reverse) {
  const fastegnn_pad_desc_t &d = tab.d[blockIdx.y];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (!reverse) {
    if (i >= (long)d.rows_dst * d.cols_dst) return;
    const int r = (int)(i / d.cols_dst), c = (int)(i % d.cols_dst);
    const int cs = r < d.rows ? pad_src_col(d, tab.h, c) : -1;
    d.dst[i] = cs >= 0 ? d.src[(size_t)r * d.cols + cs] : 0.f;
  } else {
    if (i >= (long)d.rows * d.cols) return;
    const int r = (int)(i / d.cols), c = (int)(i % d.cols);
    const_cast<float *>(d.src)[i] = d.dst[(size_t)r * d.cols_dst + pad_dst_col(d, tab.h, c)];
  }
}

__global__ void permute_rows_kernel(const float *in, const int32_t *perm, int E, int w, float *out) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)E * w) return;
  int k = (int)(idx / w), a = (int)(idx % w);
  out[idx] = in[(size_t)perm[k] * w + a];
}

__global__ void build_batch_kernel(const int64_t *b64, int N, int B, int32_t *batch, int32_t *gptr) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < N) batch[idx] = (int32_t)b64[idx];
  if (idx <= B) {
    int lo = 0, hi = N;  // first node with batch >= idx
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (b64[mid] < idx) lo = mid + 1; else hi = mid;
    }
    gptr[idx] = lo;
  }
}

// ---------------------------------------------------------------- dW += G^T T  (K = rows)
// A stage queues its 64x64 weight-gradient contractions as jobs (WgradBatch, kernels.h).  One launch
// contracts all of them: every workgroup owns a row range of one (job, batch) pair and writes its
// 64x64 partial as a plain 16 KB slab; a second small kernel sums the slabs in a fixed order and
// adds them to the gradient tensors.  No float atomics: deterministic, and no same-address
// contention (4096 atomics x hundreds of workgroups onto one 16 KB tile cost ~50 us per launch).
// The kernel streams its operands at ~4.0 TB/s against 5.3 TB/s for a bare read loop on this part
// (fastegnn_selftest_stream): it is HBM-bound, and a bf16x3 version of the inner product (6x fewer
// MFMA issue cycles) measured the same in round 1 (6.0 ms per step) and again in round 2 on the bundle geometry
// (2.93 vs 2.91 ms per step) -- the fp32-input MFMA form is kept.
constexpr int WTS = 80;  // LDS row stride of the staged operand tiles (conflict-free b32 column reads)
#ifndef FE_WG_OCC
#define FE_WG_OCC 2   // waves per SIMD the contraction kernels are compiled for (measured per step: 2 -> 2.84 ms, 3 -> 2.94 ms with 44-52 B of scratch, 4 -> 5.8 ms)
#endif

// rows [m_first, m1) of one (G, T) pair in steps of `step` rows, 16 rows at a time, into the wave's 64x64 accumulator
// (acc[ti][tk][r] = dW[16ti + 4q + r][16tk + i]); gt / tt: this wave's two 16 x WTS staging tiles.
// lane l loads 16 bytes of row (4s + q), columns 4i..4i+3: full 256-byte lines per row.  Addresses are a
// wave-uniform base (row m) plus loop-invariant 32-bit lane offsets: no per-load VALU math.
__device__ __forceinline__ void wg_accumulate(const float *G, const float *T, int ldg, int ldt, long m_first, long m1,
                                              int step, bool want_bias, bool rnd, float *gt, float *tt,
                                              f32x4 (&acc)[4][4], float (&bsum)[4]) {
  const int l = lane_id(), i = l & 15, q = l >> 4;
  f32x4 gv[4], tv[4];
  unsigned og[4], ot[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    og[s] = ((unsigned)(4 * s + q) * (unsigned)ldg + 4u * i) * 4u;
    ot[s] = ((unsigned)(4 * s + q) * (unsigned)ldt + 4u * i) * 4u;
  }
  auto issue = [&](long m) {
    const char *gb = reinterpret_cast<const char *>(G + (size_t)m * ldg);
    const char *tb = reinterpret_cast<const char *>(T + (size_t)m * ldt);
    if (m + 16 <= m1) {   // full tile (wave-uniform): unmasked loads
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        gv[s] = *reinterpret_cast<const f32x4 *>(gb + og[s]);
        tv[s] = *reinterpret_cast<const f32x4 *>(tb + ot[s]);
      }
    } else {              // ragged tail: rows beyond m1 read row m (always valid) and are zeroed below
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = m + 4 * s + q < m1;
        gv[s] = *reinterpret_cast<const f32x4 *>(gb + (ok ? og[s] : 4u * i * 4u));
        tv[s] = *reinterpret_cast<const f32x4 *>(tb + (ok ? ot[s] : 4u * i * 4u));
      }
    }
  };
  long m = m_first;
  if (m < m1) issue(m);
  for (; m < m1; m += step) {
    __builtin_amdgcn_wave_barrier();
    if (m + 16 <= m1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        *reinterpret_cast<f32x4 *>(gt + (4 * s + q) * WTS + 4 * i) = gv[s];
        *reinterpret_cast<f32x4 *>(tt + (4 * s + q) * WTS + 4 * i) = tv[s];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = m + 4 * s + q < m1;
        *reinterpret_cast<f32x4 *>(gt + (4 * s + q) * WTS + 4 * i) = ok ? gv[s] : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(tt + (4 * s + q) * WTS + 4 * i) = ok ? tv[s] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (m + step < m1) issue(m + step);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float av[4], bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        av[t] = gt[(4 * s + q) * WTS + 16 * t + i];
        bv[t] = tt[(4 * s + q) * WTS + 16 * t + i];
      }
      if (rnd) {   // bf16 operand mode (wave-uniform): products of bf16 values are exact in fp32-input MFMA
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (want_bias) bsum[t] += av[t];   // the bias gradient sums the unrounded rows
          av[t] = round_bf(av[t]);
          bv[t] = round_bf(bv[t]);
        }
      }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        if (want_bias && !rnd) bsum[ti] += av[ti];
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ti], bv[tk], acc[ti][tk], 0, 0, 0);
      }
    }
  }
}

// The same contraction with the MFMA fragments loaded straight from global memory (lane l = i + 16 k reads element
// (row m + 4 s + k, column 16 t + i): exactly the 16x16x4 operand layout, 64-byte segments of four rows per instruction), no
// LDS staging: nothing to write, read back or fence, and DEPTH tiles of loads in flight per wave instead of one.
#ifndef FE_WG_DEPTH
#define FE_WG_DEPTH 2
#endif
__device__ __forceinline__ void wg_accumulate_direct(const float *G, const float *T, int ldg, int ldt, long m_first, long m1,
                                                     int step, bool want_bias, bool rnd, f32x4 (&acc)[4][4], float (&bsum)[4]) {
  constexpr int D = FE_WG_DEPTH;
  const int l = lane_id(), i = l & 15, k = l >> 4;
  float gv[D][16], tv[D][16];
  unsigned og[4], ot[4];   // byte offsets of (row 4 s + k, column i); + 64 t bytes per column tile
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    og[s] = ((unsigned)(4 * s + k) * (unsigned)ldg + (unsigned)i) * 4u;
    ot[s] = ((unsigned)(4 * s + k) * (unsigned)ldt + (unsigned)i) * 4u;
  }
  auto issue = [&](int slot, long m) {
    const char *gb = reinterpret_cast<const char *>(G + (size_t)m * ldg);
    const char *tb = reinterpret_cast<const char *>(T + (size_t)m * ldt);
    const bool full = m + 16 <= m1;   // wave-uniform; a ragged tail reads row m (always valid) and is zeroed at use
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = full || m + 4 * s + k < m1;
      const unsigned a = ok ? og[s] : (unsigned)i * 4u, b = ok ? ot[s] : (unsigned)i * 4u;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        gv[slot][4 * s + t] = *reinterpret_cast<const float *>(gb + a + 64u * t);
        tv[slot][4 * s + t] = *reinterpret_cast<const float *>(tb + b + 64u * t);
      }
    }
  };
  auto consume = [&](int slot, long m) {
    const bool full = m + 16 <= m1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float av[4], bv[4];
      const bool ok = full || m + 4 * s + k < m1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        av[t] = ok ? gv[slot][4 * s + t] : 0.f;
        bv[t] = ok ? tv[slot][4 * s + t] : 0.f;
      }
      if (rnd) {   // bf16 operand mode (wave-uniform): products of bf16 values are exact in fp32-input MFMA
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (want_bias) bsum[t] += av[t];   // the bias gradient sums the unrounded rows
          av[t] = round_bf(av[t]);
          bv[t] = round_bf(bv[t]);
        }
      }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        if (want_bias && !rnd) bsum[ti] += av[ti];
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ti], bv[tk], acc[ti][tk], 0, 0, 0);
      }
    }
  };
  long m = m_first;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (m + (long)d * step < m1) issue(d, m + (long)d * step);
  while (m < m1) {
#pragma unroll
    for (int d = 0; d < D; ++d) {   // static slots: the ring is the unrolled loop body
      if (m < m1) {
        consume(d, m);
        if (m + (long)D * step < m1) issue(d, m + (long)D * step);
        m += step;
      }
    }
  }
}

// The contraction on the bf16 matrix pipe: 32 rows per step, fragments straight from global memory in the 16x16x32 operand
// layout (lane l = i + 16 q holds rows m + 8 q .. + 7 of column 16 t + i: eight 64-byte-segment loads per fragment), each
// fp32 operand split in registers into bf16 parts h | m | l (part_pack, as the stage kernels' activations) and six
// products per (ti, tk) tile -- 96 matrix instructions of 16 cycles per 32 rows against 128 of 32 cycles on the fp32-input
// form.  BF (bf16 operand mode): both operands rounded to bf16 (RNE) and ONE product -- the mode's weight gradients on
// the bf16 instruction itself.  The next step's G fragments are requested while this step's T fragments are split, the
// next T fragment of a column tile as soon as the current one has been split.
struct WgParts { u32x4 p[3]; };
template <bool BF>
__device__ __forceinline__ WgParts wg_split8(const float (&x)[8]) {
  WgParts P;
  if constexpr (BF) {
#pragma unroll
    for (int w = 0; w < 4; ++w) P.p[0][w] = pack_rne(x[2 * w], x[2 * w + 1]);
    P.p[1] = P.p[2] = u32x4{0u, 0u, 0u, 0u};
  } else {
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = x[e];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      P.p[0][w] = part_pack(y[2 * w], y[2 * w + 1]);
      P.p[1][w] = part_pack(y[2 * w], y[2 * w + 1]);
      P.p[2][w] = part_pack(y[2 * w], y[2 * w + 1]);
    }
  }
  return P;
}
template <bool BF>
__device__ __forceinline__ void wg_accumulate_x3(const float *G, const float *T, int ldg, int ldt, long m_first, long m1,
                                                 int step, bool want_bias, f32x4 (&acc)[4][4], float (&bsum)[4]) {
  const int l = lane_id(), i = l & 15, q = l >> 4;
  float gr[4][8], tr[4][8];   // raw fragments [column tile][row e]
  unsigned og[8], ot[8];      // byte offsets of (row 8 q + e, column i); + 64 t bytes per column tile
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    og[e] = ((unsigned)(8 * q + e) * (unsigned)ldg + (unsigned)i) * 4u;
    ot[e] = ((unsigned)(8 * q + e) * (unsigned)ldt + (unsigned)i) * 4u;
  }
  // rows at or beyond m1 read row m (always valid) and are zeroed at use
  auto load_g = [&](long m) {
    const char *gb = reinterpret_cast<const char *>(G + (size_t)m * ldg);
    const bool full = m + 32 <= m1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned a = (full || m + 8 * q + e < m1) ? og[e] : (unsigned)i * 4u;
#pragma unroll
      for (int t = 0; t < 4; ++t) gr[t][e] = *reinterpret_cast<const float *>(gb + a + 64u * t);
    }
  };
  auto load_t = [&](long m, int t) {
    const char *tb = reinterpret_cast<const char *>(T + (size_t)m * ldt);
    const bool full = m + 32 <= m1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned a = (full || m + 8 * q + e < m1) ? ot[e] : (unsigned)i * 4u;
      tr[t][e] = *reinterpret_cast<const float *>(tb + a + 64u * t);
    }
  };
  long m = m_first;
  if (m < m1) {
    load_g(m);
#pragma unroll
    for (int t = 0; t < 4; ++t) load_t(m, t);
  }
  for (; m < m1; m += step) {
    const bool full = m + 32 <= m1;
    const bool more = m + step < m1;
    WgParts gp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (!full) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (m + 8 * q + e >= m1) gr[t][e] = 0.f;
      }
      if (want_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[t] += gr[t][e];   // the bias gradient sums the unrounded rows
      }
      gp[t] = wg_split8<BF>(gr[t]);
    }
    if (more) load_g(m + step);
#pragma unroll
    for (int tk = 0; tk < 4; ++tk) {
      if (!full) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (m + 8 * q + e >= m1) tr[tk][e] = 0.f;
      }
      const WgParts tp = wg_split8<BF>(tr[tk]);
      if (more) load_t(m + step, tk);
      const bf16x8 bh = __builtin_bit_cast(bf16x8, tp.p[0]), bm = __builtin_bit_cast(bf16x8, tp.p[1]),
                   bl = __builtin_bit_cast(bf16x8, tp.p[2]);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, gp[ti].p[0]);
        if constexpr (BF) {
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[ti][tk], 0, 0, 0);
        } else {
          const bf16x8 am = __builtin_bit_cast(bf16x8, gp[ti].p[1]), al = __builtin_bit_cast(bf16x8, gp[ti].p[2]);
          // smallest terms first
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[ti][tk], 0, 0, 0);
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[ti][tk], 0, 0, 0);
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[ti][tk], 0, 0, 0);
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[ti][tk], 0, 0, 0);
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[ti][tk], 0, 0, 0);
          acc[ti][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[ti][tk], 0, 0, 0);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256, FE_WG_OCC) void wgrad_tn_kernel(WgTable tab) {
  // staged form: 40 KB of staging tiles, the 64x64 reduction buffer aliases them after the main loop; default form: the
  // four waves' partial tiles (64 KB)
#if defined(FE_WG_STAGED)
  __shared__ __attribute__((aligned(16))) float smem[4 * 2 * 16 * WTS];
#else
  __shared__ __attribute__((aligned(16))) float smem[4 * IMG];
#endif
  __shared__ float redb[H];
  float *red = smem;
  // locate the job of this workgroup
  int jb = 0;
#pragma unroll 1
  while (jb + 1 < tab.n_jobs && (int)blockIdx.x >= tab.job[jb + 1].wg_begin) ++jb;
  const WgJob &a = tab.job[jb];
  const int local = blockIdx.x - a.wg_begin;
  // batch index varies fastest: co-resident workgroups read the same row range of every batch slice
  int bidx = local % a.nb, split = local / a.nb;
#ifdef FE_WG_XCD
  // ... and the nb workgroups of one row range sit on ONE XCD (workgroup i runs on XCD i % 8): the G rows they share are
  // then served by that XCD's L2 instead of being fetched once per XCD
  if (a.nb > 1 && (a.nsplit & 7) == 0 && (a.wg_begin & 7) == 0) {
    const int xcd = local & 7, slot = local >> 3;
    bidx = slot % a.nb;
    split = xcd + 8 * (slot / a.nb);
  }
#endif
  const int l = lane_id(), i = l & 15, q = l >> 4, w = wave_id();
  const float *G = a.G + (size_t)bidx * a.sG;
  const float *T = a.T + (size_t)bidx * a.sT;
  const long m0 = (long)split * a.rows_per_wg;
  long m1 = m0 + a.rows_per_wg;
  if (m1 > a.M) m1 = a.M;
  if (threadIdx.x < H) redb[threadIdx.x] = 0.f;
  float *gt = smem + (w * 2 + 0) * 16 * WTS, *tt = smem + (w * 2 + 1) * 16 * WTS;
  f32x4 acc[4][4];
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tk = 0; tk < 4; ++tk) acc[ti][tk] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_bias = a.db != nullptr;
#if defined(FE_WG_STAGED)
  wg_accumulate(G, T, a.ldg, a.ldt, m0 + 16 * w, m1, 64, want_bias, a.round != 0, gt, tt, acc, bsum);
#elif defined(FE_WG_F32)
  (void)gt; (void)tt;
  wg_accumulate_direct(G, T, a.ldg, a.ldt, m0 + 16 * w, m1, 64, want_bias, a.round != 0, acc, bsum);
#else
  (void)gt; (void)tt;
  if (a.round) wg_accumulate_x3<true>(G, T, a.ldg, a.ldt, m0 + 32 * w, m1, 128, want_bias, acc, bsum);
  else wg_accumulate_x3<false>(G, T, a.ldg, a.ldt, m0 + 32 * w, m1, 128, want_bias, acc, bsum);
#endif
#if defined(FE_WG_STAGED) || defined(FE_WG_ATOMIC_EPILOGUE)
  __syncthreads();   // all waves are done with their staging tiles
  for (int k = threadIdx.x; k < IMG; k += 256) red[k] = 0.f;
  __syncthreads();
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tk = 0; tk < 4; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[(16 * ti + 4 * q + r) * H + 16 * tk + i], acc[ti][tk][r]);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    float s = qsum(bsum[ti]);
    if (q == 0) atomicAdd(&redb[16 * ti + i], s);
  }
  __syncthreads();
  // slab order: [job slabs][batch][split] so that the reducer walks splits contiguously
  const size_t sidx = (size_t)a.slab_begin + (size_t)bidx * a.nsplit + split;
  f32x4 *dst = reinterpret_cast<f32x4 *>(tab.slab + sidx * IMG);
  const f32x4 *src = reinterpret_cast<const f32x4 *>(red);
  for (int k = threadIdx.x; k < IMG / 4; k += 256) dst[k] = src[k];
  if (threadIdx.x < H) tab.slab_b[sidx * H + threadIdx.x] = redb[threadIdx.x];
#else
  // The four waves' 64x64 partials are added in a FIXED order (wave 0 + 1 + 2 + 3): every wave parks its accumulator in
  // LDS in register order (sixteen 16-byte stores per lane, conflict-free), then thread t sums fragment tiles 4 (t / 64) ..
  // + 3 of lane t % 64 and stores them into the slab.  (Until late round 3 every lane added its 256 values to one shared
  // tile with LDS float atomics -- 1 024 64-lane atomics per workgroup, a third of the kernel's wave time by the counters,
  // and an order-dependent sum.)
  (void)red;
  f32x4 *part = reinterpret_cast<f32x4 *>(smem) + (size_t)w * (IMG / 4);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tk = 0; tk < 4; ++tk) part[(ti * 4 + tk) * 64 + l] = acc[ti][tk];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    float s = qsum(bsum[ti]);
    if (q == 0) atomicAdd(&redb[16 * ti + i], s);   // 64 values per wave: the bias sums keep their atomics
  }
  __syncthreads();
  // slab order: [job slabs][batch][split] so that the reducer walks splits contiguously
  const size_t sidx = (size_t)a.slab_begin + (size_t)bidx * a.nsplit + split;
  float *dst = tab.slab + sidx * IMG;
  const f32x4 *p0 = reinterpret_cast<const f32x4 *>(smem);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = w * 4 + u, ti = k >> 2, tk = k & 3;
    const f32x4 v = ((p0[k * 64 + l] + p0[IMG / 4 + k * 64 + l]) + p0[2 * (IMG / 4) + k * 64 + l]) + p0[3 * (IMG / 4) + k * 64 + l];
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(16 * ti + 4 * q + r) * H + 16 * tk + i] = v[r];
  }
  if (threadIdx.x < H) tab.slab_b[sidx * H + threadIdx.x] = redb[threadIdx.x];
#endif
}

// Sum the partial slabs of every (job, batch) and accumulate into the gradients.  A workgroup owns
// 64 consecutive elements of one 64x64 tile; its 8 waves sum interleaved subsets of the splits
// (8 independent loads in flight per thread) and the 8 partials are added in a fixed order.
__global__ __launch_bounds__(512) void wgrad_reduce_kernel(WgTable tab) {
  // The partial slabs are added in DOUBLE precision (round 4): a gradient is the sum of 128-512 partials that cancel -- the
  // layer-0 bias columns of the virtual coordinate heads to ~1e-3 of their terms -- and an fp32 chain over them was the
  // largest single contribution to the excess over the reference's own error on those tensors (tests/helpers.py,
  // GRAD_EXCEPTIONS).  The kernel reads each partial once either way: 0.14 ms per step at cfg4 before and after.
  __shared__ double part[8][H];
  const WgJob &a = tab.job[blockIdx.x];
  const int bidx = blockIdx.y;
  // batches that all add into the same dW (sW == 0): their slabs are contiguous and reduced together
  const bool merged = a.nb > 1 && a.sW == 0;
  if (bidx >= (merged ? 1 : a.nb)) return;
  const int n_part = merged ? a.nsplit * a.nb : a.nsplit;
  const size_t s0 = (size_t)a.slab_begin + (size_t)bidx * a.nsplit;
  float *dW = a.dW + (size_t)bidx * a.sW;
  const int e = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const bool bias_block = blockIdx.z == IMG / H;      // last z-block reduces the bias slabs
  // (a slab job may keep its slabs in an array of the caller: add_slabs_ext stores its base in the unused operand pointer)
  const bool ext = a.T == nullptr && a.G != nullptr;
  const float *slabs = ext ? a.G : tab.slab;
  const size_t sx = ext ? (size_t)bidx * a.nsplit : s0;
  // (ext slabs come in the accumulator order of the 32x32 MFMA blocks -- common.h, WgAcc32.  The 64 threads of a z-block read 64
  //  CONSECUTIVE floats of that order and decode which (o, k) each of them is: float ((((bo 2 + bk) 4 + e4) 64 + lane) 4 + r)
  //  holds element (32 bo + 8 e4 + 4 (lane >> 5) + r, 32 bk + (lane & 31)))
  const size_t eoff = (size_t)blockIdx.z * H + e;
  int o_ = blockIdx.z, k_ = e;
  if (ext && !bias_block) {
    const int idx = (int)eoff, ln = (idx >> 2) & 63;
    o_ = 32 * (idx >> 11) + 8 * ((idx >> 8) & 3) + 4 * (ln >> 5) + (idx & 3);
    k_ = 32 * ((idx >> 10) & 1) + (ln & 31);
  }
  const float *base = bias_block ? tab.slab_b + s0 * H + e : slabs + sx * IMG + eoff;
  const size_t stride = bias_block ? H : IMG;
  if (bias_block && !a.db) return;
  double s[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  int p = pl;
  for (; p + 56 < n_part; p += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(p + 8 * u) * stride];   // eight independent loads in flight
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += (double)v[u];
  }
  for (; p < n_part; p += 8) s[0] += (double)base[(size_t)p * stride];
  part[pl][e] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (pl == 0) {
    double t = 0.;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += part[u][e];
    // (atomic: two jobs of one batch may add into the same gradient -- the sharded caller runs a layer's edge backward as two
    // launches, each with its own partial slabs; a plain read-modify-write lost one of the two sums)
    if (bias_block) {
      atomicAdd(&a.db[e], (float)t);
    } else {
      const int o = o_, k = k_;
      if (k < a.kmax) atomicAdd(dW + (size_t)o * a.lddw + a.c0 + (size_t)k * a.ks, (float)t);
    }
  }
}

WgradBatch::WgradBatch(float *slab, hipStream_t st_, bool round_bf16, int slab_base_, int slab_cap_)
    : st(st_), round(round_bf16), slab_base(slab_base_), slab_cap(slab_cap_) {
  tab.n_jobs = 0;
  min_rows = 256;
  // most workgroups (= partial slabs) per job: sweep at cfg4 (contraction + reduction, ms per step) 256: 3.22, 512: 3.17,
  // 768: 3.23, 1024: 3.28 (FE_WG_CAP overrides it for such sweeps; re-swept in round 3: 256 / 512 within 2 %, 128 and 1024 worse)
  static const int cap_default = getenv("FE_WG_CAP") ? atoi(getenv("FE_WG_CAP")) : 512;
  max_split = cap_default;
  tab.slab = slab;
  tab.slab_b = slab ? slab + (size_t)WG_SLABS * IMG : nullptr;
  n_wg = 0;
  n_slab = slab_base;
  slab_top = slab_base + slab_cap;
  planned = false;
  max_nb = 1;
}

// Slab bookkeeping: jobs whose slabs the caller's own kernel writes (add_slabs) need their slab range at once and take
// it from the TOP of the batch's share; contraction jobs (add) only record the split they would like -- their ranges are
// assigned by plan() (from finish()), scaled down together if the share cannot hold them all (many long
// jobs in one layer-wide batch: N >= 393 k with B*C >= 30 k overflowed the round-2 budget and failed the backward).
int WgradBatch::add(const float *G, int ldg, const float *T, int ldt, long M, float *dW, int lddw, int c0, int ks,
                    float *db, int nb, long sG, long sT, long sW, int kmax) {
  if (M <= 0 || !dW) return FASTEGNN_OK;
  FE_REQUIRE(G && T, "wgrad: null operand");
  FE_REQUIRE(tab.slab, "wgrad: wg_slab workspace is null");
  FE_REQUIRE(tab.n_jobs < WG_MAX_JOBS, "wgrad: too many jobs in one batch");
  FE_REQUIRE((ldg % 4) == 0 && (ldt % 4) == 0, "wgrad: operand rows must be 16-byte aligned");
  FE_REQUIRE(!planned, "wgrad: add() after the batch has been planned");
  long nsplit = (M + 1023) / 1024;            // 1024 rows per workgroup while that fills the chip ...
  // ... short operands: down to min_rows per workgroup, up to `fill` workgroups per job.  A layer's batch holds ~8 jobs of
  // N rows, so 128 per job already put 4 workgroups on every CU: measured at cfg4 (weight-gradient kernels per step)
  // fill 64: 2.96 ms, 128: 2.91 ms, 256: 3.25 ms, 512: 3.57 ms (FE_WG_FILL overrides it for such sweeps)
  static const long fill = getenv("FE_WG_FILL") ? atol(getenv("FE_WG_FILL")) : 128;
  if (nsplit < fill) nsplit = (M + min_rows - 1) / min_rows;
  if (nsplit > fill && M < fill * 1024) nsplit = fill;
  long cap = max_split / nb;
  if (cap < 4) cap = 4;
  if (nsplit > cap) nsplit = cap;
  WgJob &j = tab.job[tab.n_jobs++];
  j.G = G; j.T = T; j.dW = dW; j.db = db; j.M = M; j.sG = sG; j.sT = sT; j.sW = sW;
  j.ldg = ldg; j.ldt = ldt; j.lddw = lddw; j.c0 = c0; j.ks = ks; j.kmax = kmax;
  j.nsplit = (int)nsplit; j.nb = nb; j.rows_per_wg = 0;
  j.wg_begin = -1; j.slab_begin = -1;
  j.round = round ? 1 : 0;
  if (nb > max_nb) max_nb = nb;
  return FASTEGNN_OK;
}

// assign row ranges, workgroups and slab ranges to the contraction jobs (idempotent)
int WgradBatch::plan() {
  if (planned) return FASTEGNN_OK;
  long want = 0;
  for (int k = 0; k < tab.n_jobs; ++k)
    if (tab.job[k].T) want += (long)tab.job[k].nsplit * tab.job[k].nb;   // (T == null: a slab job, whose slabs its caller's kernel wrote)
  const long avail = (long)slab_top - slab_base;
  long need_min = 0;
  for (int k = 0; k < tab.n_jobs; ++k)
    if (tab.job[k].T) need_min += tab.job[k].nb;
  FE_REQUIRE(need_min <= avail, "wgrad: slab workspace exhausted");
  n_wg = 0;
  n_slab = slab_base;
  for (int k = 0; k < tab.n_jobs; ++k) {
    WgJob &j = tab.job[k];
    if (!j.T) { j.wg_begin = n_wg; continue; }   // slab job: contributes no workgroups to wgrad_tn_kernel
    long nsplit = j.nsplit;
    if (want > avail) {   // every job gives up the same fraction (at least one slab per batch slice stays)
      nsplit = nsplit * (avail - need_min) / want;
      if (nsplit < 1) nsplit = 1;
    }
    long rows = (j.M + nsplit - 1) / nsplit;
    rows = (rows + 63) / 64 * 64;
    nsplit = (j.M + rows - 1) / rows;
    j.rows_per_wg = (int)rows;
    j.nsplit = (int)nsplit;
    j.wg_begin = n_wg;
    j.slab_begin = n_slab;
    n_wg += (int)(nsplit * j.nb);
    n_slab += (int)(nsplit * j.nb);
  }
  FE_REQUIRE(n_slab <= slab_top, "wgrad: slab workspace exhausted");
  planned = true;
  return FASTEGNN_OK;
}

int WgradBatch::guard_write(const float *p, size_t n, const char *what) const {
  if (!p || n == 0) return FASTEGNN_OK;
  const char *w0 = reinterpret_cast<const char *>(p), *w1 = w0 + n * sizeof(float);
  for (int k = 0; k < tab.n_jobs; ++k) {
    const WgJob &j = tab.job[k];
    if (!j.G || j.M <= 0) continue;   // slab jobs have no operands
    const size_t span_g = (size_t)(j.nb - 1) * (size_t)(j.sG > 0 ? j.sG : 0) + (size_t)(j.M - 1) * j.ldg + H;
    const size_t span_t = (size_t)(j.nb - 1) * (size_t)(j.sT > 0 ? j.sT : 0) + (size_t)(j.M - 1) * j.ldt + H;
    const char *g0 = reinterpret_cast<const char *>(j.G), *t0 = reinterpret_cast<const char *>(j.T);
    const bool hit = (w0 < g0 + span_g * sizeof(float) && g0 < w1) || (w0 < t0 + span_t * sizeof(float) && t0 < w1);
    if (hit) {
      set_error(std::string("wgrad: a stage is about to overwrite ") + what + " while queued weight-gradient job " +
                std::to_string(k) + " still has to read it (the contractions run at finish())");
      return FASTEGNN_E_INVALID;
    }
  }
  return FASTEGNN_OK;
}

int WgradBatch::add_slabs(float *dW, int lddw, int c0, int ks, float *db, int nsplit, int *slab_begin) {
  FE_REQUIRE(dW && slab_begin && nsplit > 0, "wgrad: add_slabs arguments");
  FE_REQUIRE(tab.slab, "wgrad: wg_slab workspace is null");
  FE_REQUIRE(tab.n_jobs < WG_MAX_JOBS, "wgrad: too many jobs in one batch");
  FE_REQUIRE(!planned, "wgrad: add_slabs() after the batch has been planned");
  FE_REQUIRE(slab_top - nsplit >= slab_base, "wgrad: slab workspace exhausted");
  slab_top -= nsplit;                         // from the top of the share: the range is needed now, by the caller's kernel
  WgJob &j = tab.job[tab.n_jobs++];
  j.G = nullptr; j.T = nullptr; j.dW = dW; j.db = db; j.M = 0; j.sG = j.sT = j.sW = 0;
  j.ldg = j.ldt = H; j.lddw = lddw; j.c0 = c0; j.ks = ks; j.kmax = 64;
  j.rows_per_wg = 0; j.nsplit = nsplit; j.nb = 1;
  j.round = 0;
  j.wg_begin = -1; j.slab_begin = slab_top;
  *slab_begin = slab_top;
  return FASTEGNN_OK;
}

int WgradBatch::add_slabs_ext(const float *ext, float *dW, int lddw, int c0, int ks, int nsplit, int nb, long sW) {
  FE_REQUIRE(ext && dW && nsplit > 0 && nb > 0 && sW != 0, "wgrad: add_slabs_ext arguments");
  FE_REQUIRE(tab.n_jobs < WG_MAX_JOBS, "wgrad: too many jobs in one batch");
  FE_REQUIRE(!planned, "wgrad: add_slabs_ext() after the batch has been planned");
  WgJob &j = tab.job[tab.n_jobs++];
  j.G = ext; j.T = nullptr; j.dW = dW; j.db = nullptr; j.M = 0; j.sG = j.sT = 0; j.sW = sW;
  j.ldg = j.ldt = H; j.lddw = lddw; j.c0 = c0; j.ks = ks; j.kmax = 64;
  j.rows_per_wg = 0; j.nsplit = nsplit; j.nb = nb;
  j.round = 0;
  j.wg_begin = -1; j.slab_begin = 0;
  if (nb > max_nb) max_nb = nb;
  return FASTEGNN_OK;
}

int WgradBatch::finish() {
  if (tab.n_jobs == 0) return FASTEGNN_OK;
  int rc = plan();
  if (rc) return rc;
  if (n_wg > 0) {
    { ProfScope _ps(K_WGRAD_TN, st); hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, tab); }
    rc = check_launch("wgrad_tn_kernel");
    if (rc) return rc;
  }
  { ProfScope _ps(K_WGRAD_REDUCE, st); hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)tab.n_jobs, (unsigned)max_nb, IMG / H + 1), dim3(512), 0, st, tab); }
  tab.n_jobs = 0;   // the batch may be refilled: slabs and workgroup ranges start over
  n_wg = 0;
  n_slab = slab_base;
  slab_top = slab_base + slab_cap;
  planned = false;
  max_nb = 1;
  return check_launch("wgrad_reduce_kernel");
}

// ---------------------------------------------------------------- dW[:, c0+a] += G^T F, a < kf <= 8
struct WgsArgs {
  const float *G, *F;
  float *dW, *db;
  long M;
  int ldg, ldf, kf, lddw, c0, rows_per_wg;
};
// (round 4: double accumulators -- per thread, in the workgroup's LDS sums and therefore in the partial that leaves the
// workgroup -- and at most 256 workgroups (64 measured 0.27 instead of 0.05 ms per step at cfg4: too few rows in flight): the gradient of embedding_in is the end of the whole backward chain, a column sum
// over N rows that cancels; with fp32 chains and up to 512 float atomics in arrival order it measured 2.4 x the reference's
// own error on the goldens.  The final += into dW / db is still a float atomic per workgroup: <= 256 well-rounded partials.)
__global__ __launch_bounds__(256) void wgrad_small_kernel(WgsArgs a) {
  const int o = threadIdx.x & 63, w = wave_id();
  const long m0 = (long)blockIdx.x * a.rows_per_wg;
  long m1 = m0 + a.rows_per_wg;
  if (m1 > a.M) m1 = a.M;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double bs = 0.;
  long m = m0 + w;
  for (; m + 12 < m1; m += 16) {   // four rows in flight per wave
    float g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) g[u] = a.G[(size_t)(m + 4 * u) * a.ldg + o];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bs += (double)g[u];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < a.kf) acc[k] += (double)g[u] * (double)a.F[(size_t)(m + 4 * u) * a.ldf + k];
    }
  }
  for (; m < m1; m += 4) {
    const float g = a.G[(size_t)m * a.ldg + o];
    bs += (double)g;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < a.kf) acc[k] += (double)g * (double)a.F[(size_t)m * a.ldf + k];
  }
  __shared__ double red[4][9][H];
#pragma unroll
  for (int k = 0; k < 8; ++k) red[w][k][o] = acc[k];
  red[w][8][o] = bs;
  __syncthreads();
  if (w == 0) {   // the four waves' partials in a fixed order
    for (int k = 0; k < a.kf; ++k)
      atomicAdd(&a.dW[(size_t)o * a.lddw + a.c0 + k], (float)((red[0][k][o] + red[1][k][o]) + (red[2][k][o] + red[3][k][o])));
    if (a.db) atomicAdd(&a.db[o], (float)((red[0][8][o] + red[1][8][o]) + (red[2][8][o] + red[3][8][o])));
  }
}
static int launch_wgrad_small_b(const float *G, int ldg, const float *F, int ldf, int kf, long M, float *dW, int lddw,
                                int c0, float *db, hipStream_t st) {
  if (M <= 0 || !dW) return FASTEGNN_OK;
  FE_REQUIRE(kf >= 0 && kf <= 8, "wgrad_small: kf > 8 unsupported");
  WgsArgs a{G, F, dW, db, M, ldg, ldf, kf, lddw, c0, 0};
  long nsplit = (M + 255) / 256;
  if (nsplit > 256) nsplit = 256;
  long rows = (M + nsplit - 1) / nsplit;
  a.rows_per_wg = (int)rows;
  nsplit = (M + rows - 1) / rows;
  { ProfScope _ps_wgrad_small_kernel(K_WGRAD_SMALL, st); hipLaunchKernelGGL(wgrad_small_kernel, dim3((unsigned)nsplit), dim3(256), 0, st, a); }
  return check_launch("wgrad_small_kernel");
}
int launch_dgrad_small(const float *G, long N, int kf, const float *W, int ldw, int c0, float *g_in, int accumulate,
                       hipStream_t st) {
  if (N == 0 || kf == 0) return FASTEGNN_OK;
  hipLaunchKernelGGL(dgrad_small_kernel, dim3(cdiv(N * kf, 256)), dim3(256), 0, st, G, N, kf, W, ldw, c0, g_in, accumulate);
  return check_launch("dgrad_small_kernel");
}

int launch_wgrad_small(const float *G, int ldg, const float *F, int ldf, int kf, long M, float *dW, int lddw, int c0,
                       hipStream_t st) {
  return launch_wgrad_small_b(G, ldg, F, ldf, kf, M, dW, lddw, c0, nullptr, st);
}

// HBM streaming calibration (bench/tests only): mode 0 read + reduce, 1 copy, 2 write
__global__ __launch_bounds__(256) void stream_kernel(const f32x4 *src, f32x4 *dst, size_t n, int mode) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
    for (; k + 3 * stride < n; k += 4 * stride) {
      const f32x4 a = src[k], b = src[k + stride], c = src[k + 2 * stride], d = src[k + 3 * stride];
      acc += (a + b) + (c + d);
    }
    for (; k < n; k += stride) acc += src[k];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) dst[0] = acc;   // keep the loads alive
  } else if (mode == 1) {
    for (; k < n; k += stride) dst[k] = src[k];
  } else {
    for (; k < n; k += stride) dst[k] = acc;
  }
}

// ---------------------------------------------------------------- toolkit self-test
// Y[j][o] = sum_k A[o][k] X[j][k] for one 16-item tile, A = W or W^T (64x64 row-major).
__global__ __launch_bounds__(64) void selftest_gemm_kernel(const float *W, const float *X, float *Y, int transposed) {
  __shared__ __attribute__((aligned(16))) float img[IMG];
  for (int idx = threadIdx.x; idx < IMG; idx += 64) {
    int o = idx >> 6, k = idx & 63;
    img[img_index(o, k)] = transposed ? W[k * H + o] : W[o * H + k];
  }
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4;
  Vec in = vload_row(X + j * H, q);
  Vec acc = vzero();
  gemm64(img, in, acc);
  // second hop through an identity-free chain: Y2 = A * silu(Y) is not needed; store directly
  vstore_row(Y + j * H, q, acc);
}

// Row-major split image (common.h): Y[j][.] = W X[j] (transposed = 0) or W^T X[j] (1) for one 16-item tile, in the
// bf16x3 form (mode 0) or the single-product bf16 form on rounded operands (mode 1); the image is built in LDS
// exactly as pack_kernel writes it.
__global__ __launch_bounds__(64) void selftest_rm_kernel(const float *W, const float *X, float *Y, int transposed, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned rm[RM_WORDS];
  for (int idx = threadIdx.x; idx < 64 * (RM_RS / 4); idx += 64) {
    const int o = idx / (RM_RS / 4), w = idx % (RM_RS / 4);
    float w0 = w < 32 ? W[o * H + 2 * w] : 0.f, w1 = w < 32 ? W[o * H + 2 * w + 1] : 0.f;
    if (mode == 1) { w0 = round_bf(w0); w1 = round_bf(w1); }
    for (int p = 0; p < 3; ++p) rm[p * (RM_PART / 4) + idx] = split_word(w0, w1, p);
  }
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4;
  const Vec in = vload_row(X + j * H, q);
  Vec acc = vzero();
  const char *img = reinterpret_cast<const char *>(rm);
  if (mode == 0) {
    if (transposed) gemm_rm<GM_X3, true>(img, vsplit(in), acc);
    else gemm_rm<GM_X3, false>(img, vsplit(in), acc);
  } else {
    if (transposed) gemm_rm<GM_BF16, true>(img, vpack_bf(in), acc);
    else gemm_rm<GM_BF16, false>(img, vpack_bf(in), acc);
  }
  vstore_row(Y + j * H, q, acc);
}

// jreduce16 (common.h): out[q*16 + j] = sum over the 16 items of a quarter-row q of X[item][16 (j >> 2) + 4 q + (j & 3)]
// lane l: out[l] = qsum(X[l]) (over the four lanes l % 16 + 16 q), out[64 + l] = jsum(X[l]) (over the 16 lanes of its row)
__global__ __launch_bounds__(64) void selftest_lane_sums_kernel(const float *X, float *out) {
  const float x = X[threadIdx.x];
  out[threadIdx.x] = qsum(x);
  out[64 + threadIdx.x] = jsum(x);
}
__global__ __launch_bounds__(64) void selftest_jreduce_kernel(const float *X, float *out) {
  const int l = lane_id(), j = l & 15, q = l >> 4;
  const Vec v = vload_row(X + j * H, q);
  out[l] = jreduce16(v);
}

// bf16x3 counterpart of chain_kernel (mode bit0: SiLU, bit2: single layer written to out for checks, bit3: the f16x2 form)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void chain_bf3_kernel(const float *W, const float *X, float *out, int iters,
                                                               int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds3[];
  for (int idx = threadIdx.x; idx < 3 * 2048; idx += blockDim.x) lds3[idx] = 0u;
  __syncthreads();
  // split the fp32 weights W[o][k] into the three bf16 images (what pack.hip would do once per step)
  for (int idx = threadIdx.x; idx < IMG; idx += blockDim.x) {
    const int o = idx >> 6, k = idx & 63;
    float w = W[idx], dummy = 0.f;
    const int tile = k >> 4, r = k & 3, e = ((tile & 1) << 2) | r;
    for (int p = 0; p < 3; ++p) {
      const unsigned hb = part_pack(w, dummy) & 0xffffu;   // the bf16 part of w (low half of the pair word); w <- residual
      atomicOr(&lds3[img3_index(p, o, k)], (e & 1) ? (hb << 16) : hb);
    }
  }
  __syncthreads();
  if (mode & 8) {   // f16x2 image (parts h | l) instead of the bf16 h | m | l parts
    for (int idx = threadIdx.x; idx < 3 * 2048; idx += blockDim.x) lds3[idx] = 0u;
    __syncthreads();
    for (int idx = threadIdx.x; idx < IMG / 2; idx += blockDim.x) {
      const int o = idx >> 5, k = (idx & 31) * 2;
      lds3[img3_index(0, o, k)] = split2_word(W[o * H + k], W[o * H + k + 1], 0);
      lds3[img3_index(1, o, k)] = split2_word(W[o * H + k], W[o * H + k + 1], 1);
    }
    __syncthreads();
  }
  const int l = lane_id(), j = l & 15, q = l >> 4;
  Vec x = vload_row(X + j * H, q);
  for (int it = 0; it < iters; ++it) {
    Vec acc = vzero();
    if (mode & 8) gemm64_f2(lds3, vsplit2(x), acc);
    else gemm64_x3(lds3, vsplit(x), acc);
    x = (mode & 1) ? vsilu(acc) : vscale(acc, 0.125f);
    if (mode & 4) x = acc;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) vstore_row(out + j * H, q, x);
}

// MFMA-chain micro-benchmark: `iters` dependent 64x64 layers per wave (optionally with SiLU), the
// weight image in LDS or read from global memory.  Used to calibrate what the gemm64 building block
// can reach on its own (tests/bench only).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void chain_kernel(const float *wimg, float *out, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  load_images(lds, wimg, 1);
  __syncthreads();
  const int l = lane_id(), q = l >> 4;
  const float *img = (mode & 2) ? wimg : lds;
  Vec x;
#pragma unroll
  for (int t = 0; t < 4; ++t) x.t[t] = f32x4{0.01f * l, 0.02f, -0.01f * t, 0.03f};
  for (int it = 0; it < iters; ++it) {
    Vec acc = vzero();
    gemm64(img, x, acc);
    x = (mode & 1) ? vsilu(acc) : vscale(acc, 0.125f);
  }
  if (out) vstore_row(out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64 * 64 * 0, q, x);
}

}  // namespace fe

using namespace fe;

extern "C" {

int fastegnn_embed_forward(const float *node_feat, int32_t N, int32_t nf, const float *W, const float *b, float *h,
                           void *stream) {
  FE_REQUIRE(node_feat && W && b && h, "embed_forward: null pointer");
  if (N == 0) return FASTEGNN_OK;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(cdiv((long)N * H, 256)), dim3(256), 0, (hipStream_t)stream, node_feat, N, nf,
                     W, b, h);
  return check_launch("embed_fwd_kernel");
}

int fastegnn_embed_backward(const float *node_feat, const float *g_h, int32_t N, int32_t nf, const float *W, float *gW,
                            float *gb, float *g_node_feat, void *stream) {
  FE_REQUIRE(node_feat && g_h && W, "embed_backward: null pointer");
  FE_REQUIRE(nf <= 8, "embed_backward: node_feat_nf > 8 unsupported");
  if (N == 0) return FASTEGNN_OK;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_wgrad_small_b(g_h, H, node_feat, nf, nf, N, gW, nf, 0, gb, st);
  if (rc) return rc;
  if (g_node_feat) {
    return launch_dgrad_small(g_h, N, nf, W, nf, 0, g_node_feat, 0, st);
  }
  return FASTEGNN_OK;
}

int fastegnn_virtual_init(const float *vnf, int32_t B, int32_t C, float *HvT, void *stream) {
  FE_REQUIRE(vnf && HvT, "virtual_init: null pointer");
  hipLaunchKernelGGL(virtual_init_kernel, dim3(cdiv((long)B * C * H, 256)), dim3(256), 0, (hipStream_t)stream, vnf, B, C,
                     HvT);
  return check_launch("virtual_init_kernel");
}

int fastegnn_virtual_init_backward(const float *g_HvT, int32_t B, int32_t C, float *g_vnf, void *stream) {
  FE_REQUIRE(g_HvT && g_vnf, "virtual_init_backward: null pointer");
  hipLaunchKernelGGL(virtual_init_bwd_kernel, dim3(cdiv((long)C * H, 256)), dim3(256), 0, (hipStream_t)stream, g_HvT, B,
                     C, g_vnf);
  return check_launch("virtual_init_bwd_kernel");
}

int fastegnn_pad_params(const fastegnn_pad_desc_t *desc, int32_t n, int32_t h, int32_t reverse, void *stream) {
  FE_REQUIRE(n == 0 || desc, "pad_params: null descriptors");
  FE_REQUIRE(h >= 1 && h <= fe::H, "pad_params: hidden_nf must be in [1,64]");
  for (int k0 = 0; k0 < n; k0 += fe::PAD_MAX_DESC) {
    fe::PadTable tab;
    tab.n = n - k0 < fe::PAD_MAX_DESC ? n - k0 : fe::PAD_MAX_DESC;
    tab.h = h;
    long most = 0;
    for (int k = 0; k < tab.n; ++k) {
      const fastegnn_pad_desc_t &d = desc[k0 + k];
      FE_REQUIRE(d.src && d.dst && d.rows >= 0 && d.cols >= 0 && d.rows_dst >= d.rows && d.nblk >= 0 && d.nblk <= 3 && d.lead >= 0,
                 "pad_params: bad descriptor");
      int cs = d.lead, cd = d.lead;
      for (int b = 0; b < d.nblk; ++b) {
        FE_REQUIRE(d.blk[b] >= 0 && d.blk[b] % h == 0, "pad_params: a block must be a multiple of hidden_nf columns");
        cs += d.blk[b];
        cd += d.blk[b] / h * fe::H;
      }
      FE_REQUIRE(cs <= d.cols && d.cols_dst == cd + (d.cols - cs), "pad_params: column blocks do not add up");
      tab.d[k] = d;
      const long elems = reverse ? (long)d.rows * d.cols : (long)d.rows_dst * d.cols_dst;
      if (elems > most) most = elems;
    }
    if (most == 0) continue;
    hipLaunchKernelGGL(fe::pad_params_kernel, dim3((unsigned)cdiv(most, 256), (unsigned)tab.n), dim3(256), 0, (hipStream_t)stream,
                       tab, reverse);
  }
  return check_launch("pad_params_kernel");
}

int fastegnn_permute_rows(const float *in, const int32_t *perm, int32_t E, int32_t width, float *out, void *stream) {
  if (E == 0 || width == 0) return FASTEGNN_OK;
  FE_REQUIRE(in && perm && out, "permute_rows: null pointer");
  hipLaunchKernelGGL(permute_rows_kernel, dim3(cdiv((long)E * width, 256)), dim3(256), 0, (hipStream_t)stream, in, perm, E,
                     width, out);
  return check_launch("permute_rows_kernel");
}

int fastegnn_build_batch(const int64_t *batch64, int32_t N, int32_t B, int32_t *batch, int32_t *gptr, void *stream) {
  FE_REQUIRE(batch64 && batch && gptr, "build_batch: null pointer");
  int n = N > B + 1 ? N : B + 1;
  hipLaunchKernelGGL(build_batch_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, batch64, N, B, batch, gptr);
  return check_launch("build_batch_kernel");
}

int fastegnn_selftest_gemm(const float *W, const float *X, float *Y, int32_t transposed, void *stream) {
  FE_REQUIRE(W && X && Y, "selftest_gemm: null pointer");
  hipLaunchKernelGGL(selftest_gemm_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, W, X, Y, transposed);
  return check_launch("selftest_gemm_kernel");
}

int fastegnn_selftest_lane_sums(const float *X, float *out, void *stream) {
  FE_REQUIRE(X && out, "selftest_lane_sums: null pointer");
  hipLaunchKernelGGL(selftest_lane_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, X, out);
  return check_launch("selftest_lane_sums_kernel");
}

int fastegnn_selftest_jreduce(const float *X, float *out, void *stream) {
  FE_REQUIRE(X && out, "selftest_jreduce: null pointer");
  hipLaunchKernelGGL(selftest_jreduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, X, out);
  return check_launch("selftest_jreduce_kernel");
}

int fastegnn_selftest_rm(const float *W, const float *X, float *Y, int32_t transposed, int32_t mode, void *stream) {
  FE_REQUIRE(W && X && Y, "selftest_rm: null pointer");
  hipLaunchKernelGGL(selftest_rm_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, W, X, Y, transposed, mode);
  return check_launch("selftest_rm_kernel");
}

int fastegnn_selftest_chain(const float *wimg, float *out, int32_t iters, int32_t mode, int32_t waves, int32_t grid,
                            void *stream) {
  FE_REQUIRE(wimg && out, "selftest_chain: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = IMG * sizeof(float);
  if (waves == 4) hipLaunchKernelGGL(chain_kernel<4>, dim3(grid), dim3(256), lds, st, wimg, out, iters, mode);
  else if (waves == 8) hipLaunchKernelGGL(chain_kernel<8>, dim3(grid), dim3(512), lds, st, wimg, out, iters, mode);
  else if (waves == 16) hipLaunchKernelGGL(chain_kernel<16>, dim3(grid), dim3(1024), lds, st, wimg, out, iters, mode);
  else { set_error("selftest_chain: waves must be 4, 8 or 16"); return FASTEGNN_E_INVALID; }
  return check_launch("chain_kernel");
}

int fastegnn_selftest_chain_bf3(const float *W, const float *X, float *out, int32_t iters, int32_t mode, int32_t waves,
                                int32_t grid, void *stream) {
  FE_REQUIRE(W && X && out, "selftest_chain_bf3: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 3 * 2048 * sizeof(unsigned);
  if (waves == 4) hipLaunchKernelGGL(chain_bf3_kernel<4>, dim3(grid), dim3(256), lds, st, W, X, out, iters, mode);
  else if (waves == 8) hipLaunchKernelGGL(chain_bf3_kernel<8>, dim3(grid), dim3(512), lds, st, W, X, out, iters, mode);
  else if (waves == 16) hipLaunchKernelGGL(chain_bf3_kernel<16>, dim3(grid), dim3(1024), lds, st, W, X, out, iters, mode);
  else { set_error("selftest_chain_bf3: waves must be 4, 8 or 16"); return FASTEGNN_E_INVALID; }
  return check_launch("chain_bf3_kernel");
}

int fastegnn_selftest_stream(const float *src, float *dst, size_t n_floats, int32_t mode, void *stream) {
  FE_REQUIRE(src && dst && n_floats % 4 == 0, "selftest_stream: bad arguments");
  hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4 *>(src),
                     reinterpret_cast<f32x4 *>(dst), n_floats / 4, mode);
  return check_launch("stream_kernel");
}

// Host-only: the slab planning of a weight-gradient batch (no launch, no device memory touched).  Queues n_jobs jobs of
// M[k] rows x nb[k] batch slices, then n_slab_jobs caller-written slab jobs of `slabs_each` slabs, into a batch with the
// given split limit and slab share; nsplit_out[k] receives the partial slabs per batch slice of job k.
int fastegnn_selftest_wgrad_plan(const int64_t *M, const int32_t *nb, int32_t n_jobs, int32_t n_slab_jobs, int32_t slabs_each,
                                 int32_t max_split, int32_t slab_cap, int32_t *nsplit_out) {
  FE_REQUIRE(M && nb && nsplit_out && n_jobs >= 0, "selftest_wgrad_plan: bad arguments");
  static float dummy[4];
  WgradBatch wb(dummy, nullptr, false, 0, slab_cap);
  wb.max_split = max_split;
  for (int k = 0; k < n_jobs; ++k) {
    int rc = wb.add(dummy, H, dummy, H, (long)M[k], dummy, H, 0, 1, nullptr, nb[k], 0, 0, 0);
    if (rc) return rc;
  }
  for (int k = 0; k < n_slab_jobs; ++k) {
    int first = 0;
    int rc = wb.add_slabs(dummy, H, 0, 1, nullptr, slabs_each, &first);
    if (rc) return rc;
  }
  int rc = wb.plan();
  if (rc) return rc;
  for (int k = 0; k < n_jobs; ++k) nsplit_out[k] = wb.tab.job[k].nsplit;
  FE_REQUIRE(wb.n_slab <= wb.slab_top, "selftest_wgrad_plan: slab budget exceeded");
  return FASTEGNN_OK;
}

// Host-only: the overwrite guard of an open batch.  A job reading M rows of 64 floats at `base` (G) and at
// base + 64 M (T) is queued; returns guard_write(base + probe_off, probe_n): 0 = disjoint, FASTEGNN_E_INVALID = overlap.
int fastegnn_selftest_wgrad_guard(int64_t M, int64_t probe_off, int64_t probe_n) {
  static float arena[4];
  WgradBatch wb(arena, nullptr, false, 0, 64);
  const float *base = reinterpret_cast<const float *>((uintptr_t)1 << 32);   // never dereferenced
  int rc = wb.add(base, H, base + M * H, H, (long)M, arena, H, 0, 1, nullptr);
  if (rc) return rc;
  return wb.guard_write(base + probe_off, (size_t)probe_n, "the probe range");
}

int fastegnn_selftest_wgrad(const float *G, const float *T, int32_t M, float *dW, float *db, float *slab,
                            void *stream) {
  WgradBatch wb(slab, (hipStream_t)stream);
  int rc = wb.add(G, H, T, H, M, dW, H, 0, 1, db, 1, 0, 0, 0);
  if (rc) return rc;
  return wb.finish();
}

}  // extern "C"
