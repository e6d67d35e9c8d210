// The two GEMM kernels of the WIDE path (csrc/wide.hip): every nn.Linear of a hidden_nf > 64 model (models/FastEGNN.py:28-99),
// its input gradient and its weight gradient.  fp32 in, fp32 out, fp32-grade products on the bf16 matrix pipe: each operand is
// split x = h + m + l into three bf16 values (part_pack, common.h) and a product is the six terms hh, hm, mh, hl, lh, mm on
// v_mfma_f32_32x32x16_bf16 (the dropped ml, lm, ll terms are below 2^-26 relative; bf16 carries fp32's exponent range, so
// nothing here can overflow that the reference's fp32 would not).  Six bf16 products cost 6/16 of one fp32 MFMA product.
//
// gemm_x3_kernel   C[m, n] = epi( sum_k pro(A[m, k]) * B(k, n) )        rows m stream, B (a weight block) stays in LDS
//   workgroup = 8 waves; B's panel (<= 128 k x 32 NQ columns) is split once into an LDS image in MFMA-fragment order and reused
//   for every row unit the workgroup walks; a wave owns 32 rows x 32 NQ columns, stages its own 32 x 32 fp32 chunk of A through a
//   private LDS strip (full 128-byte lines from HBM, no workgroup barrier), reads 8 consecutive k per lane back, applies the
//   optional activation (prologue: A is a pre-activation), splits in registers.
//   epilogue: + bias[n] + base[m, n], * act'(Z[m, n]) (the activation's backward fused into the input gradient), accumulate.
// tn_x3_kernel     dW[o, k] += sum_m G[m, o] * pro(X[m, k])              both operands stream, the contraction runs over rows
//   workgroup = 4 waves on a 128 x 128 block of dW; 32-row tiles of G and X are split at staging into row-major bf16 images
//   ([m][column], 320-byte rows) and read TRANSPOSED with ds_read_b64_tr_b16 (conflict-free at that stride); fp32 atomics into
//   dW per row range as before; the bias gradient rides along in the staging threads' registers.
#pragma once
#include <type_traits>
#include "kernels.h"

namespace fe {
namespace wide {

typedef float f32x16w __attribute__((ext_vector_type(16)));

constexpr int ACT_NONE = -1;
// how a kernel instance treats its fused activation: none, SiLU inline (the reference's default act_fn), or any kind through ONE
// out-of-line function (the generic switch inlined per element multiplied the GEMM kernel's code by 20 and its time by 2)
// AM_HEAD_*: (GEMM prologue only) the A operand is GENERATED: A[m, k] = gs[m] * w2[k] * act'(Zc[m, k]) from the stored
// pre-activation Zc of a scalar head  s = act(X W1^T + b1) . w2^T  (coord_mlp_* / gravity_mlp, models/FastEGNN.py:55-99) and the
// head's output gradient gs -- the gradient of Zc is never stored.
// AM_DOT_*: (GEMM epilogue only) besides C = the head's hidden pre-activation, the head's OUTPUT s[m] = act(C[m, :]) . w2 + b2 is
// formed from the accumulators (one column block: N <= 128) -- the second Linear of the head costs no pass over C.
enum { AM_NONE = 0, AM_SILU = 1, AM_GEN = 2, AM_HEAD_SILU = 3, AM_HEAD_GEN = 4, AM_DOT_SILU = 5, AM_DOT_GEN = 6 };
inline int am_of(int kind) { return kind < 0 ? AM_NONE : kind == FASTEGNN_ACT_SILU ? AM_SILU : AM_GEN; }
__device__ __noinline__ float act_gen(float z, int kind, float p) { return act_f(z, Act{kind, p}); }
__device__ __noinline__ float dact_gen(float z, int kind, float p) { return dact_f(z, Act{kind, p}); }
template <int AM>
__device__ __forceinline__ float pro_t(float z, Act a) {
  if constexpr (AM == AM_NONE) return z;
  else if constexpr (AM == AM_SILU) return silu_f(z);
  else return act_gen(z, a.kind, a.p);
}
// y = act(z), d = act'(z)
template <int AM>
__device__ __forceinline__ void both_t(float z, Act a, float &y, float &d) {
  if constexpr (AM == AM_SILU) silu_both(z, y, d);
  else {
    y = act_gen(z, a.kind, a.p);
    d = dact_gen(z, a.kind, a.p);
  }
}
template <int AM>
__device__ __forceinline__ float dact_t(float z, Act a) {
  if constexpr (AM == AM_NONE) return 1.f;
  else if constexpr (AM == AM_SILU) {
    float y, d;
    silu_both(z, y, d);
    return d;
  } else return dact_gen(z, a.kind, a.p);
}

constexpr int XKP = 128;                 // contraction panel held in LDS
constexpr int XWAVES = 8;                // waves per workgroup of gemm_x3_kernel
constexpr int XRS = 144;                 // bytes per row of a wave's A strip: 32 fp32 + 16 (rows 9 slots apart: conflict-free ds_read_b128)
constexpr int XSTRIP = 32 * XRS;         // 4 608 bytes per wave
__host__ __device__ constexpr int x3_b_bytes(int nq) { return 3 * (XKP / 16) * nq * 64 * 16; }
__host__ __device__ constexpr int x3_lds_bytes(int nq) { return x3_b_bytes(nq) + XWAVES * XSTRIP; }

struct GemmX3 {
  const float *A; int lda; long M; int Kd;
  const float *Bp; long sbk, sbn; int N;
  const float *bias, *base; float *C; int ldc;
  int accumulate;
  Act pro;                               // A := act(A) when pro.kind >= 0
  const float *Z; int ldz; Act epi;      // C *= act'(Z[m, n]) when Z
  int units_per_wave;                    // 32-row units each wave walks (uniform: the panels' barriers are workgroup-wide)
  const float *gs, *w2;                  // AM_HEAD_*: A := gs[m] * w2[k] * act'(A[m, k]);  AM_DOT_*: w2 = the head's second weight
  float *sout; const float *b2;          // AM_DOT_*: sout[m] = act(C[m, :]) . w2 + b2[0]
};

#ifdef FE_WIDE_GEMM_IMPL
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// three bf16 parts of 8 fp32 values as MFMA operands
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
  float y[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = x[e];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    h[w] = part_pack(y[2 * w], y[2 * w + 1]);
    m[w] = part_pack(y[2 * w], y[2 * w + 1]);
    l[w] = part_pack(y[2 * w], y[2 * w + 1]);
  }
}
__device__ __forceinline__ void mma6(const u32x4 &ah, const u32x4 &am, const u32x4 &al, const u32x4 &bh, const u32x4 &bm,
                                     const u32x4 &bl, f32x16w &c) {
  const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
  const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, c, 0, 0, 0);   // small terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, c, 0, 0, 0);
}

// The wave's work is ONE stream of k16 steps (unit after unit, panel after panel, chunk after chunk, two steps per chunk), software-
// pipelined by hand and pinned with sched_group_barrier:
//   step s issues   [odd steps: the next chunk's strip write, the global loads of the chunk after it]
//                   the A read of step s + 1, then per column quadrant: the B fragment reads of the NEXT quadrant, the six MFMAs of
//                   this one, and -- in the MFMAs' shadow -- a quarter of step s + 1's activation and bf16 split.
// The compiler's own order (read, wait, MFMA, read, wait ...) exposed an LDS round trip in front of every pair of MFMAs and the
// whole split between steps.  Rows beyond M and columns beyond K are read from clamped (valid) addresses: rows beyond M are never
// stored, columns beyond K meet B's zero rows -- no control flow inside the stream.  Requires lda % 4 == 0 and a 16-byte
// aligned A (the host routes other shapes to gemm_smallk_kernel).
// DEEP: four chunk buffers instead of one -- a wave keeps 16 KB of A in flight (a whole 128-wide unit ahead) instead of 4 KB; with one
// buffer the 8 waves of a compute unit had 32 KB outstanding, a quarter of what HBM's latency asks for, and every chunk waited.
// The four buffers rotate through a chunk loop unrolled by four with fixed names (a rotation by index or by copy made the compiler
// move freshly loaded registers, i.e. wait for them), so DEEP runs every panel as four chunks (the host picks it when the padding of
// the last panel costs less than a third).
template <int NQ, int PRO, int EPI, bool DEEP>
__global__ __launch_bounds__(XWAVES * 64) void gemm_x3_kernel(GemmX3 g) {
  extern __shared__ __attribute__((aligned(16))) char x3_smem[];
  unsigned *bimg = reinterpret_cast<unsigned *>(x3_smem);
  constexpr int KS = XKP / 16;                       // k16 steps of a full panel
  constexpr int PART_WORDS = KS * NQ * 256;          // u32 words of one part of the B image
  constexpr int VPER = PRO == AM_NONE ? 3 : (PRO == AM_SILU ? 6 : (PRO == AM_HEAD_SILU ? 8 : 3));   // vector instructions scheduled behind each MFMA
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: so is all unit addressing)
  const int l32 = lane & 31, hh = lane >> 5;
  char *strip = x3_smem + x3_b_bytes(NQ) + wave * XSTRIP;
  const int n0 = blockIdx.y * (32 * NQ);
  const int npanels = (g.Kd + XKP - 1) / XKP;
  // chunks per unit (DEEP: every panel as four chunks -- a contraction that is not a multiple of 128 wide runs its last panel
  // padded: the columns beyond K re-read valid ones against B's zero rows)
  const int cpu = DEEP ? ((g.Kd + XKP - 1) / XKP) * 4 : (g.Kd + 31) >> 5;
  const int lr = lane >> 3, lc = lane & 7;            // loader: rows lr + 8 i, float4 column lc of a 32 x 32 chunk
  const long wave_id = (long)blockIdx.x * XWAVES + wave, wave_n = (long)gridDim.x * XWAVES;
  const u32x4 *bfr = reinterpret_cast<const u32x4 *>(bimg) + lane;
  const char *aread = strip + l32 * XRS + 32 * hh;
  char *awrite = strip + lr * XRS + 16 * lc;
  constexpr bool HEAD = PRO >= AM_HEAD_SILU;
  constexpr int DACT = PRO == AM_HEAD_SILU ? AM_SILU : AM_GEN;
  // HEAD: w2 over the whole contraction (zero beyond K: the padded columns then contribute nothing) behind the strips, and the
  // head gradient of the lane's row for this unit and the next (the split of a unit's first step is made in the last step of the
  // unit before it)
  float *w2s = reinterpret_cast<float *>(x3_smem + x3_b_bytes(NQ) + XWAVES * XSTRIP);
  float gsc = 0.f, gsn = 0.f;
  auto gs_of = [&](int unit) {
    long m = (wave_id + (long)unit * wave_n) * 32 + l32;
    return g.gs[m < g.M ? m : g.M - 1];
  };
  if constexpr (HEAD) {
    for (int k = tid; k < cpu * 32; k += XWAVES * 64) w2s[k] = k < g.Kd ? g.w2[k] : 0.f;
    gsn = gs_of(0);
    __syncthreads();
  }

  // ---- the prefetch cursor: chunk pw of unit pu
  int pu = 0, pw = 0;
  // (a unit's base address is scalar, the lane's part a 32-bit offset: 64-bit per-lane addresses cost the registers that spilled)
  // (columns beyond K re-read the row's last four valid columns: B's rows beyond K are zero, and a select on the loaded value would
  //  wait for the load wherever the scheduler puts it -- inside the stream it serialised the prefetch)
  struct Chunk { float4 v0, v1, v2, v3; };   // (by value: as arrays behind lambda references the two buffers stayed in scratch memory)
  auto load_next = [&]() {
    long mu = (wave_id + (long)pu * wave_n) * 32;
    mu = mu < g.M ? mu : g.M - 1;
    const float *ub = g.A + (size_t)mu * g.lda;
    const long left = g.M - 1 - mu;
    const int lim = left < 31 ? (int)left : 31;       // last valid row of the unit
    int k = 32 * pw + 4 * lc;
    k = k >= g.Kd ? g.Kd - 4 : k;
    auto row = [&](int i) {
      const int rr = lr + 8 * i < lim ? lr + 8 * i : lim;
      return *reinterpret_cast<const float4 *>(ub + (unsigned)(rr * g.lda + k));
    };
    Chunk r{row(0), row(1), row(2), row(3)};
    if (++pw == cpu) { pw = 0; ++pu; }
    return r;
  };
  auto store_strip = [&](const Chunk &r) {
    *reinterpret_cast<float4 *>(awrite) = r.v0;
    *reinterpret_cast<float4 *>(awrite + 8 * XRS) = r.v1;
    *reinterpret_cast<float4 *>(awrite + 16 * XRS) = r.v2;
    *reinterpret_cast<float4 *>(awrite + 24 * XRS) = r.v3;
  };
  // activation + bf16 split of word w (two of the lane's eight values) of a step
  // (gsv, kn -- HEAD only: the row's head gradient and the first column of the step the split is made for)
  auto split_word = [&](const float4 &x0, const float4 &x1, int w, u32x4 &h, u32x4 &m, u32x4 &l, float gsv, int kn) {
    float a = w == 0 ? x0.x : w == 1 ? x0.z : w == 2 ? x1.x : x1.z;
    float b = w == 0 ? x0.y : w == 1 ? x0.w : w == 2 ? x1.y : x1.w;
    if constexpr (HEAD) {
      const float2 wv = *reinterpret_cast<const float2 *>(w2s + kn + 8 * hh + 2 * w);
      a = gsv * wv.x * dact_t<DACT>(a, g.pro);
      b = gsv * wv.y * dact_t<DACT>(b, g.pro);
    } else if constexpr (PRO != AM_NONE) {
      // (zero padding beyond K: act(0) may be nonzero, but B's rows beyond K are zero)
      a = pro_t<PRO>(a, g.pro);
      b = pro_t<PRO>(b, g.pro);
    }
    h[w] = part_pack(a, b);
    m[w] = part_pack(a, b);
    l[w] = part_pack(a, b);
  };

  Chunk R0, R1, R2, R3;   // (R0, R2, R3: DEEP only)
  if constexpr (DEEP) {
    R0 = load_next();
    R1 = load_next();
    R2 = load_next();
    R3 = load_next();
    store_strip(R0);
    R0 = load_next();
  } else {
    R1 = load_next();
    store_strip(R1);
    R1 = load_next();
  }
  u32x4 ah, am, al;
  {
    const float4 x0 = *reinterpret_cast<const float4 *>(aread), x1 = *reinterpret_cast<const float4 *>(aread + 16);
#pragma unroll
    for (int w = 0; w < 4; ++w) split_word(x0, x1, w, ah, am, al, gsn, 0);
  }
  u32x4 bh = bfr[0], bm = bfr[PART_WORDS / 4], bl = bfr[2 * (PART_WORDS / 4)];   // (garbage until the first image is built: reloaded below)

  for (int it = 0; it < g.units_per_wave; ++it) {
    const long m0 = (wave_id + (long)it * wave_n) * 32;   // may lie beyond M: the wave then computes on clamped rows and stores nothing
    f32x16w acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if constexpr (HEAD) {
      gsc = gsn;
      gsn = gs_of(it + 1);
    }
    for (int pn = 0; pn < npanels; ++pn) {
      const int kp0 = pn * XKP;
      const int klen = g.Kd - kp0 < XKP ? g.Kd - kp0 : XKP;
      const int nchunks = DEEP ? 4 : (klen + 31) >> 5;
      if (npanels > 1 || it == 0) {
        // ---- B image of the panel: word (part, k16 step s, quadrant q, lane, w) = bf16 pair k = kp0 + 16 s + 8 hh + 2 w, + 1 of column n0 + 32 q + l32
        if (npanels > 1) __syncthreads();
        for (int idx = tid; idx < 2 * nchunks * NQ * 256; idx += XWAVES * 64) {
          const int w = idx & 3, ln = (idx >> 2) & 63, rest = idx >> 8;
          const int q = rest % NQ, s = rest / NQ;
          const int k = kp0 + 16 * s + 8 * (ln >> 5) + 2 * w, n = n0 + 32 * q + (ln & 31);
          float a0 = 0.f, a1 = 0.f;
          if (n < g.N) {
            if (k < g.Kd) a0 = g.Bp[(size_t)k * g.sbk + (size_t)n * g.sbn];
            if (k + 1 < g.Kd) a1 = g.Bp[(size_t)(k + 1) * g.sbk + (size_t)n * g.sbn];
          }
          const int o = (s * NQ + q) * 256 + ln * 4 + w;
          bimg[o] = part_pack(a0, a1);
          bimg[PART_WORDS + o] = part_pack(a0, a1);
          bimg[2 * PART_WORDS + o] = part_pack(a0, a1);
        }
        __syncthreads();
        bh = bfr[0]; bm = bfr[PART_WORDS / 4]; bl = bfr[2 * (PART_WORDS / 4)];
      }
      const int last_step = 2 * nchunks - 1;
      // one k16 step; ODD: the second step of its chunk (the strip is free after the A read of this step was issued one step ago)
      // MODE 0: first step of a chunk; 1 (2, 3, 4): second step -- the next chunk goes from R1 (R2, R3, R0) to the strip and the
      // buffer takes the newest load
      auto step = [&](int s, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool ODD = MODE != 0;
        if constexpr (MODE == 1) {
          store_strip(R1);
          R1 = load_next();
        } else if constexpr (MODE == 2) {
          store_strip(R2);
          R2 = load_next();
        } else if constexpr (MODE == 3) {
          store_strip(R3);
          R3 = load_next();
        } else if constexpr (MODE == 4) {
          store_strip(R0);
          R0 = load_next();
        }
        const char *ap = aread + (ODD ? 0 : 64);
        const float4 x0 = *reinterpret_cast<const float4 *>(ap), x1 = *reinterpret_cast<const float4 *>(ap + 16);
        const int sn = s < last_step ? s + 1 : 0;    // the next step's fragments (a following panel rebuilds the image and reloads)
        float gsv = 0.f;
        int kn = 0;
        if constexpr (HEAD) {
          const bool unit_end = s == last_step && pn == npanels - 1;
          kn = s < last_step ? kp0 + 16 * (s + 1) : (unit_end ? 0 : kp0 + XKP);
          gsv = unit_end ? gsn : gsc;
        }
        u32x4 nh, nm, nl, th, tm, tl;
        static_for<0, NQ>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const u32x4 *np = bfr + ((q + 1 < NQ ? s : sn) * NQ + (q + 1 < NQ ? q + 1 : 0)) * 64;
          th = np[0]; tm = np[PART_WORDS / 4]; tl = np[2 * (PART_WORDS / 4)];
          mma6(ah, am, al, bh, bm, bl, acc[q]);
          bh = th; bm = tm; bl = tl;
#pragma unroll
          for (int w = 4 * q / NQ; w < 4 * (q + 1) / NQ; ++w) split_word(x0, x1, w, nh, nm, nl, gsv, kn);
          if constexpr (q == 0 && ODD) {
            __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);   // the strip writes, then the chunk's four global loads, first
            __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, q == 0 ? 5 : 3, 0);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPER * 4 / NQ > 0 ? VPER * 4 / NQ : 1, 0);
          }
        });
        ah = nh; am = nm; al = nl;
      };
      if constexpr (DEEP) {
        for (int c = 0; c < nchunks; c += 4) {   // (every panel of a DEEP contraction has 4 chunks)
          step(2 * c, std::integral_constant<int, 0>{});
          step(2 * c + 1, std::integral_constant<int, 1>{});
          step(2 * c + 2, std::integral_constant<int, 0>{});
          step(2 * c + 3, std::integral_constant<int, 2>{});
          step(2 * c + 4, std::integral_constant<int, 0>{});
          step(2 * c + 5, std::integral_constant<int, 3>{});
          step(2 * c + 6, std::integral_constant<int, 0>{});
          step(2 * c + 7, std::integral_constant<int, 4>{});
        }
      } else {
        for (int c = 0; c < nchunks; ++c) {
          step(2 * c, std::integral_constant<int, 0>{});
          step(2 * c + 1, std::integral_constant<int, 1>{});
        }
      }
    }
    // ---- epilogue: acc[q][r] is row 8 (r / 4) + 4 hh + r % 4, column 32 q + l32 of the unit
    if (m0 < g.M) {
      const int rows = g.M - m0 < 32 ? (int)(g.M - m0) : 32;
      const float *bsrc = g.base ? g.base : (g.accumulate ? g.C : nullptr);
      float *cu = g.C + (size_t)m0 * g.ldc;
      const float *bu = bsrc ? bsrc + (size_t)m0 * g.ldc : nullptr;
      const float *zu = (EPI == AM_SILU || EPI == AM_GEN) ? g.Z + (size_t)m0 * g.ldz : nullptr;
      // (opaque per unit: the 64 x 2 element offsets below are invariant across units, and hoisted out of the unit loop they
      //  were spilled -- 300 registers of scratch traffic whose vmcnt(0) waits also drained the A prefetch inside the stream)
      int ldc = g.ldc, ldz = g.ldz;
      asm volatile("" : "+v"(ldc), "+v"(ldz));
      // FULL: a whole unit inside a whole column block -- no per-element predicate (the guarded form is a branch per element).
      // The addend is `base` (forward) or the old C (accumulate) -- the host never asks for both -- and comes after the activation
      // factor; every load of a quadrant goes first (C may be the addend: the compiler cannot move a load above an earlier store).
      constexpr bool DOT = EPI >= AM_DOT_SILU;
      constexpr int DOTACT = EPI == AM_DOT_SILU ? AM_SILU : AM_GEN;
      constexpr bool MUL = EPI == AM_SILU || EPI == AM_GEN;
      auto emit = [&](auto full_tag, auto addend_tag) {
        constexpr bool FULL = decltype(full_tag)::value, ADD = decltype(addend_tag)::value;
        float sp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sp[r] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int n = n0 + 32 * q + l32;
          if (!FULL && n >= g.N) continue;
          const float bn = g.bias ? g.bias[n] : 0.f;
          float w2n = 0.f;
          if constexpr (DOT) w2n = g.w2[n];
          float bv[16], zv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * hh + (r & 3);
            const bool ok = FULL || row < rows;
            if constexpr (ADD) bv[r] = ok ? bu[(unsigned)(row * ldc + n)] : 0.f;
            else bv[r] = 0.f;
            if constexpr (MUL) zv[r] = ok ? zu[(unsigned)(row * ldz + n)] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * hh + (r & 3);
            if (!FULL && row >= rows) continue;
            float v = acc[q][r] + bn;
            if constexpr (MUL) v *= dact_t<EPI>(zv[r], g.epi);
            v += bv[r];
            cu[(unsigned)(row * ldc + n)] = v;
            if constexpr (DOT) sp[r] += pro_t<DOTACT>(v, g.epi) * w2n;
          }
        }
        if constexpr (DOT) {   // the row sums over the 32 lanes of the half-wave that holds the row
          const float b2v = g.b2 ? g.b2[0] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float t = sp[r];
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) t += __shfl_xor(t, d, 32);
            const int row = 8 * (r >> 2) + 4 * hh + (r & 3);
            if (l32 == 0 && (FULL || row < rows)) g.sout[m0 + row] = t + b2v;
          }
        }
      };
      const bool full = rows == 32 && n0 + 32 * NQ <= g.N;
      if (full) {
        if (bu) emit(std::true_type{}, std::true_type{});
        else emit(std::true_type{}, std::false_type{});
      } else {
        if (bu) emit(std::false_type{}, std::true_type{});
        else emit(std::false_type{}, std::false_type{});
      }
    }
  }
}

#endif  // FE_WIDE_GEMM_IMPL

// ---- dW[o, c0 + k] += sum_m G[m, o] * pro(X[m, k]) over the workgroup's row range --------------------------------------------------
constexpr int TB = 128;                    // block of dW: TB outputs x TB inputs
constexpr int TRS = 320;                   // bytes per row of a staged part: 128 bf16 + 64 (4 rows of a transposed read land 16 banks apart)
constexpr int TPART = 32 * TRS;            // 32 rows
constexpr int TN_LDS = 6 * TPART;          // G parts h|m|l, X parts h|m|l: 61 440 bytes

struct TnX3 {
  const float *G; int ldg; const float *X; int ldx; long M; int O, Kd;
  float *dW; int ldw, c0; long rows_per_split; float *db;
  Act pro;                                 // X := act(X) when pro.kind >= 0
  const float *gs, *w2; Act gen;           // GEN: G[m, o] := gs[m] * w2[o] * act'(G[m, o])  (the head form, see AM_HEAD_*)
  float *dw2;                              // GEN: dw2[o] += sum_m gs[m] * act(G[m, o]), the head's second weight gradient (may be null)
};

#ifdef FE_WIDE_GEMM_IMPL
template <int PRO, int GEN>
__global__ __launch_bounds__(256) void tn_x3_kernel(TnX3 t) {
  __shared__ __attribute__((aligned(16))) char sm[TN_LDS];
  __shared__ double bred[8][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = blockIdx.x * TB, k0 = blockIdx.y * TB;
  const long r_lo = (long)blockIdx.z * t.rows_per_split, r_hi = r_lo + t.rows_per_split < t.M ? r_lo + t.rows_per_split : t.M;
  const int wo = (wave >> 1) * 64, wk = (wave & 1) * 64;     // the wave's 64 x 64 quarter of the block
  const bool do_bias = t.db != nullptr && blockIdx.y == 0;
  const bool vg = (t.ldg & 3) == 0 && (reinterpret_cast<size_t>(t.G) & 15) == 0;
  const bool vx = (t.ldx & 3) == 0 && (reinterpret_cast<size_t>(t.X) & 15) == 0;
  const int sr = tid >> 5, sc = tid & 31;                   // staging: rows sr + 8 i, float4 column sc
  f32x16w acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  double bs[4] = {0.0, 0.0, 0.0, 0.0}, b2s[4] = {0.0, 0.0, 0.0, 0.0};
  const bool do_w2 = GEN != AM_NONE && t.dw2 != nullptr && blockIdx.y == 0;
  auto load4 = [&](const float *P, int ld, bool vec, long m, int c, int ncols) {
    float4 v = float4{0.f, 0.f, 0.f, 0.f};
    if (m < r_hi && c < ncols) {
      const float *p = P + (size_t)m * ld + c;
      if (vec && c + 3 < ncols) v = *reinterpret_cast<const float4 *>(p);
      else {
        v.x = p[0];
        if (c + 1 < ncols) v.y = p[1];
        if (c + 2 < ncols) v.z = p[2];
        if (c + 3 < ncols) v.w = p[3];
      }
    }
    return v;
  };
  auto load_tile = [&](long r0, float4 (&gv)[4], float4 (&xv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gv[i] = load4(t.G, t.ldg, vg, r0 + sr + 8 * i, o0 + 4 * sc, t.O);
      xv[i] = load4(t.X, t.ldx, vx, r0 + sr + 8 * i, k0 + 4 * sc, t.Kd);
    }
  };
  auto put = [&](char *base, int row, float4 v) {   // the three parts of four values -> 8 bytes each
    float y[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      u32x2 w;
      w[0] = part_pack(y[0], y[1]);
      w[1] = part_pack(y[2], y[3]);
      *reinterpret_cast<u32x2 *>(base + p * TPART + row * TRS + 8 * sc) = w;
    }
  };
  // transposed fragment: lane (group gq = lane >> 4, i = lane & 15) receives column cb + 16 (gq & 1) + i of rows 16 s + 8 (gq >> 1) + 0..7
  const int gq = lane >> 4, li = lane & 15;
  const int tr_off = (8 * (gq >> 1) + (li >> 2)) * TRS + (16 * (gq & 1) + 4 * (li & 3)) * 2;
  auto frag = [&](const char *img, int part, int s, int cb) {
    const char *p = img + part * TPART + 16 * s * TRS + cb * 2 + tr_off;
    const u32x2 lo = lds_tr_read(p), hi = lds_tr_read(p + 4 * TRS);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  char *Gs = sm, *Xs = sm + 3 * TPART;
  float4 gv[4], xv[4];
  float gsr[4] = {0.f, 0.f, 0.f, 0.f};     // GEN: the head gradient of the staged rows
  float4 w2v = float4{0.f, 0.f, 0.f, 0.f};
  if constexpr (GEN != AM_NONE) {
    const int o = o0 + 4 * sc;
    w2v = float4{o < t.O ? t.w2[o] : 0.f, o + 1 < t.O ? t.w2[o + 1] : 0.f, o + 2 < t.O ? t.w2[o + 2] : 0.f, o + 3 < t.O ? t.w2[o + 3] : 0.f};
  }
  auto load_gs = [&](long r0) {
    if constexpr (GEN != AM_NONE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) gsr[i] = r0 + sr + 8 * i < r_hi ? t.gs[r0 + sr + 8 * i] : 0.f;
    }
  };
  load_tile(r_lo, gv, xv);
  load_gs(r_lo);
  for (long r0 = r_lo; r0 < r_hi; r0 += 32) {
    __syncthreads();            // the previous tile's reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (GEN != AM_NONE) {   // (columns beyond O: w2 = 0; rows beyond the range: gs = 0)
        float y[4], d[4];
        both_t<GEN>(gv[i].x, t.gen, y[0], d[0]); both_t<GEN>(gv[i].y, t.gen, y[1], d[1]);
        both_t<GEN>(gv[i].z, t.gen, y[2], d[2]); both_t<GEN>(gv[i].w, t.gen, y[3], d[3]);
        gv[i].x = gsr[i] * w2v.x * d[0]; gv[i].y = gsr[i] * w2v.y * d[1];
        gv[i].z = gsr[i] * w2v.z * d[2]; gv[i].w = gsr[i] * w2v.w * d[3];
        if (do_w2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) b2s[e] += (double)(gsr[i] * y[e]);
        }
      }
      if (do_bias) {
        bs[0] += (double)gv[i].x; bs[1] += (double)gv[i].y; bs[2] += (double)gv[i].z; bs[3] += (double)gv[i].w;
      }
      float4 x = xv[i];
      if constexpr (PRO != AM_NONE) {
        // (rows beyond the range were loaded as zeros and act(0) may be nonzero: G's zero rows cancel them)
        x.x = pro_t<PRO>(x.x, t.pro); x.y = pro_t<PRO>(x.y, t.pro); x.z = pro_t<PRO>(x.z, t.pro); x.w = pro_t<PRO>(x.w, t.pro);
      }
      put(Gs, sr + 8 * i, gv[i]);
      put(Xs, sr + 8 * i, x);
    }
    __syncthreads();
    if (r0 + 32 < r_hi) {
      load_tile(r0 + 32, gv, xv);
      load_gs(r0 + 32);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 a[2][3], b[2][3];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[h2][p] = frag(Gs, p, s, wo + 32 * h2);
          b[h2][p] = frag(Xs, p, s, wk + 32 * h2);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma6(a[i][0], a[i][1], a[i][2], b[j][0], b[j][1], b[j][2], acc[i][j]);
    }
  }
  // acc[i][j][r]: output o0 + wo + 32 i + 8 (r / 4) + 4 hh + r % 4, input k0 + wk + 32 j + l32
  const int l32 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk + 32 * j + l32;
      if (k >= t.Kd) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wo + 32 * i + 8 * (r >> 2) + 4 * hh + (r & 3);
        if (o < t.O) atomicAdd(t.dW + (size_t)o * t.ldw + t.c0 + k, acc[i][j][r]);
      }
    }
  if (do_bias) {   // column 4 sc + e summed over the 8 staging rows' threads
#pragma unroll
    for (int e = 0; e < 4; ++e) bred[sr][4 * sc + e] = bs[e];
    __syncthreads();
    if (tid < 128 && o0 + tid < t.O) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += bred[r][tid];
      atomicAdd(t.db + o0 + tid, (float)s);
    }
  }
  if (do_w2) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) bred[sr][4 * sc + e] = b2s[e];
    __syncthreads();
    if (tid < 128 && o0 + tid < t.O) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += bred[r][tid];
      atomicAdd(t.dw2 + o0 + tid, (float)s);
    }
  }
}

#endif  // FE_WIDE_GEMM_IMPL

// host side of the two kernels (wide_gemm.hip -- its own translation unit: 25 instances, two minutes of compile time)
int launch_gemm_x3(const GemmX3 &g, int nq, int pro_mode, int epi_mode, bool deep, dim3 grid, hipStream_t st);
int launch_tn_x3(const TnX3 &t, int pro_mode, int gen_mode, dim3 grid, hipStream_t st);

}  // namespace wide
}  // namespace fe
