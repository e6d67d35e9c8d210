// Wave-level toolkit shared by every FastEGNN kernel (gfx950 / CDNA4, wave64).
//
// Layout convention ("D layout"): a wave works on a tile of 16 items (edges, nodes or
// (graph,channel) rows).  lane = 16*q + j, j = item in the tile, q = 0..3.  A 64-wide hidden
// vector of item j is held by the four lanes (j,0..3) as 4 x f32x4:
//        Vec.t[t][r]  ==  element  16*t + 4*q + r
// This is exactly the C/D fragment map of v_mfma_f32_16x16x4_f32 when the GEMM is oriented
//        D[out][item] = sum_k W[out][k] * X[k][item]           (weights = A operand, items = B)
// so the output of one 64x64 layer is already the B operand of the next one: MLP chains
// never leave registers.  The weight matrix is read as a pre-permuted "image" (img_index)
// so that each lane fetches its A operands for four k-steps with one 16-byte load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64;
constexpr int QXLD = 68;      // Q[64] | x[3] | pad
constexpr int TS = 68;        // row stride (floats) of the 16x64 transpose tiles in LDS
constexpr int IMG = 4096;     // floats per 64x64 weight image
constexpr int FEATW = 8;      // row stride of the per-edge scalar-feature rows (r | edge_attr)

struct Vec {
  f32x4 t[4];
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// wave index in the workgroup / in the grid, as scalars: everything derived from them (chunk bounds,
// row pointers, loop trip counts) then lives in SGPRs, is fetched with scalar loads and branches on scc
// instead of exec masks
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int global_wave_id() { return (int)(blockIdx.x * (blockDim.x >> 6)) + wave_id(); }

// ---- activations (fp32; v_exp_f32 / v_rcp_f32 are ~1 ulp) ----
__device__ __forceinline__ float sigmoid_f(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }
__device__ __forceinline__ float dsilu_f(float z) {
  float s = sigmoid_f(z);
  return s * (1.0f + z * (1.0f - s));
}
// silu(z) and its derivative from ONE sigmoid: d = s*(1 + z*(1-s)) = s + y*(1-s) with y = z*s
__device__ __forceinline__ void silu_both(float z, float &y, float &d) {
#if defined(FE_SIGMOID_NEWTON) || defined(FE_EXP_ACCURATE)   // measured levers (round-2 verdict item 7), backward recompute only
#ifdef FE_EXP_ACCURATE   // exp(-z) with the rounding error of the exp2 argument corrected (two fma + one fma)
  const float t = -z * 1.44269504f;
  const float lo = fmaf(-z, 1.44269504f, -t) + (-z) * 1.92596299e-8f;
  const float e0 = __builtin_amdgcn_exp2f(t);
  const float den = 1.0f + fmaf(e0, lo * 0.69314718f, e0);
#else
  const float den = 1.0f + __expf(-z);
#endif
  float s = __builtin_amdgcn_rcpf(den);
#ifdef FE_SIGMOID_NEWTON   // one Newton step on the reciprocal
  s = fmaf(fmaf(-den, s, 1.0f), s, s);
#endif
#else
  const float s = sigmoid_f(z);
#endif
  y = z * s;
  d = s + y * (1.0f - s);
}
__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt_f(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float tanh_f(float z) {
  // tanh(z) = 2*sigmoid(2z) - 1 cancels for small |z| (absolute error ~1e-7, i.e. 1e-4 relative at |z| = 1e-3 -- and the
  // coordinate heads this is applied to are initialised with gain 1e-3): below 0.25 the odd Taylor polynomial to z^9
  // (relative error < 3e-7) is used instead.  One value per item, so the cost does not matter.
  const float z2 = z * z;
  const float p = z * (1.0f + z2 * (-0.33333334f + z2 * (0.13333334f + z2 * (-0.053968254f + z2 * 0.021869488f))));
  const float s = 2.0f * sigmoid_f(2.0f * z) - 1.0f;
  return fabsf(z) < 0.25f ? p : s;
}

__device__ __forceinline__ Vec vzero() {
  Vec v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v.t[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  return v;
}
template <typename F>
__device__ __forceinline__ Vec vmap(const Vec &a, F f) {
  Vec o;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) o.t[t][r] = f(a.t[t][r]);
  return o;
}
template <typename F>
__device__ __forceinline__ Vec vmap2(const Vec &a, const Vec &b, F f) {
  Vec o;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) o.t[t][r] = f(a.t[t][r], b.t[t][r]);
  return o;
}
__device__ __forceinline__ Vec vsilu(const Vec &a) { return vmap(a, [](float z) { return silu_f(z); }); }
// y = silu(z); z is overwritten by silu'(z)  (the backward keeps the derivative instead of the pre-activation)
__device__ __forceinline__ Vec vsilu_keep_d(Vec &z) {
  Vec y;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float yy, dd;
      silu_both(z.t[t][r], yy, dd);
      y.t[t][r] = yy;
      z.t[t][r] = dd;
    }
  return y;
}
// ---- activation functions other than SiLU (act_fn of the reference constructor, models/FastEGNN.py:227) ----
// Only the library built with -DFE_ACT_GENERIC (libfastegnn_hip_act.so) carries them: there every activation site takes
// an Act (kind from the FASTEGNN_F_ACT bits of the layer flags, parameter from fastegnn_layer_t.act_param) through the
// FE_ACT(args) macro, which expands to nothing in the default build -- the SiLU kernels are untouched.
struct Act {
  int kind;      // FASTEGNN_ACT_*
  float p;       // negative_slope (LeakyReLU), alpha (ELU), beta (Softplus)
};
#ifdef FE_ACT_GENERIC
#define FE_ACT(a) , fe::Act{((a).flags >> FASTEGNN_F_ACT_SHIFT) & FASTEGNN_F_ACT_MASK, (a).act_param}
#define FE_ACT_P , Act act
#define FE_ACT_A , act
__device__ __forceinline__ float expm1_f(float z) {   // exp(z) - 1 without cancellation near 0
  const float p = z * (1.0f + z * (0.5f + z * (0.16666667f + z * (0.041666668f + z * (0.008333334f + z * 0.0013888889f)))));
  return fabsf(z) < 0.3f ? p : __expf(z) - 1.0f;
}
// y = f(z), d = f'(z)
__device__ __forceinline__ void act_both(float z, Act a, float &y, float &d) {
  switch (a.kind) {
    // (every kind propagates NaN like torch does -- fmaxf / fminf would launder a NaN into a finite value, and the range guard of the
    //  f16x2 build, fastegnn_check_finite, reads the OUTPUTS: an overflow must stay visible there)
    case FASTEGNN_ACT_RELU: y = z < 0.f ? 0.f : z; d = z > 0.f ? 1.f : 0.f; break;
    case FASTEGNN_ACT_LEAKY_RELU: y = z > 0.f ? z : a.p * z; d = z > 0.f ? 1.f : a.p; break;
    case FASTEGNN_ACT_TANH: y = tanh_f(z); d = 1.0f - y * y; break;
    case FASTEGNN_ACT_SIGMOID: y = sigmoid_f(z); d = y * (1.0f - y); break;
    case FASTEGNN_ACT_ELU: { const float e = a.p * expm1_f(z < 0.f ? z : 0.f); y = z <= 0.f ? e : z; d = z > 0.f ? 1.f : e + a.p; break; }
    case FASTEGNN_ACT_GELU: {
      const float c = 0.5f * (1.0f + erff(z * 0.70710678f));
      y = z * c;
      d = c + z * 0.3989422804f * __expf(-0.5f * z * z);
      break;
    }
    case FASTEGNN_ACT_SOFTPLUS: {   // torch.nn.Softplus(beta, threshold = 20)
      const float bz = a.p * z;
      const float s = sigmoid_f(bz);
      y = bz <= 20.f ? log1pf(__expf(bz)) / a.p : z;
      d = bz > 20.f ? 1.f : s;
      break;
    }
    default: silu_both(z, y, d); break;
  }
}
__device__ __forceinline__ float act_f(float z, Act a) { float y, d; act_both(z, a, y, d); return y; }
__device__ __forceinline__ float dact_f(float z, Act a) { float y, d; act_both(z, a, y, d); return d; }
__device__ __forceinline__ Vec vsilu(const Vec &v, Act a) { return vmap(v, [a](float z) { return act_f(z, a); }); }
__device__ __forceinline__ Vec vsilu_keep_d(Vec &z, Act a) {
  Vec y;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float yy, dd;
      act_both(z.t[t][r], a, yy, dd);
      y.t[t][r] = yy;
      z.t[t][r] = dd;
    }
  return y;
}
__device__ __forceinline__ float dsilu_f(float z, Act a) { return dact_f(z, a); }
#else
#define FE_ACT(a)
#define FE_ACT_P
#define FE_ACT_A
#endif
__device__ __forceinline__ Vec vmul(const Vec &a, const Vec &b) {
  return vmap2(a, b, [](float x, float y) { return x * y; });
}
__device__ __forceinline__ Vec vscale(const Vec &a, float s) { return vmap(a, [s](float z) { return z * s; }); }
// Element-wise forms: the f32x4 forms (-DFE_PACKED_VEC) lower to v_pk_fma_f32 / v_pk_add_f32, which the constants table of
// MI355X_MICROARCH.md prices above the two plain instructions they replace when MFMAs are in flight.  Measured in round 4, two
// repeats on one box: cfg4 step 10.988 -> 10.950 ms with the plain instructions (72 v_pk_fma_f32 -> 144 v_fmac_f32 in edge_fwd).
__device__ __forceinline__ void vaxpy(Vec &acc, float s, const Vec &a) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#ifndef FE_PACKED_VEC
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.t[t][r] = __builtin_fmaf(s, a.t[t][r], acc.t[t][r]);
#else
    acc.t[t] += s * a.t[t];
#endif
  }
}
__device__ __forceinline__ void vadd(Vec &acc, const Vec &a) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#ifndef FE_PACKED_VEC
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.t[t][r] += a.t[t][r];
#else
    acc.t[t] += a.t[t];
#endif
  }
}

// natural-order 64-vector (bias, head weight, ...) -> this lane's 16 elements
__device__ __forceinline__ Vec vload_vec(const float *w, int q) {
  Vec v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v.t[t] = *reinterpret_cast<const f32x4 *>(w + 16 * t + 4 * q);
  return v;
}
// row-major [*,ld] row -> D layout (16-byte loads; row base and ld must be 16-byte aligned)
__device__ __forceinline__ Vec vload_row(const float *row, int q) { return vload_vec(row, q); }
__device__ __forceinline__ void vstore_row(float *row, int q, const Vec &v) {
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(row + 16 * t + 4 * q) = v.t[t];
}

// Same with a wave-uniform base pointer and a 32-bit per-lane element offset: the compiler can then
// use the SGPR-base + VGPR-offset addressing form, so a lane carries ONE offset register for
// several arrays instead of a 64-bit pointer per array (register pressure of the fused kernels).
// (`off` counts floats and must stay below 2^30: the byte offset is formed in 32 bits.)
__device__ __forceinline__ Vec vload_u(const float *base, unsigned off) {
  Vec v;
  const unsigned bo = off * 4u;
  const char *p = reinterpret_cast<const char *>(base);
#pragma unroll
  for (int t = 0; t < 4; ++t) v.t[t] = *reinterpret_cast<const f32x4 *>(p + (bo + 64u * t));
  return v;
}
__device__ __forceinline__ void vstore_u(float *base, unsigned off, const Vec &v) {
  const unsigned bo = off * 4u;
  char *p = reinterpret_cast<char *>(base);
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(p + (bo + 64u * t)) = v.t[t];
}

// sum over the four q-lanes of an item (xor 16, xor 32)
// (gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16- / 32-lane rows of one register with the even rows of
// another: with both holding p, the sum of the two results is p[l] + p[l ^ 16] (resp. ^ 32) in EVERY lane -- one vector
// instruction + one add per step and no trip through the LDS crossbar (ds_bpermute, which is what __shfl_xor compiles to).
// Same additions in the same order as the shuffle form: bitwise identical.  Inline assembly: the builtin's second result is
// mis-modelled by this compiler (tools/scratch/permlane_test.hip pins the semantics on the GPU).  The partner lanes of a
// q-sum hold the same item, so they are active together; -DFE_QSUM_SHFL restores the shuffles.)
__device__ __forceinline__ float qsum(float p) {
#ifdef FE_QSUM_SHFL
  p += __shfl_xor(p, 16);
  p += __shfl_xor(p, 32);
  return p;
#else
  float a = p, b = p;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  p = a + b;
  a = p; b = p;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
#endif
}
// sum over the 16 items of a tile that share q: every lane of the 16-lane row gets the total.  DPP row rotations (pure
// vector instructions; all lanes of the row must be active); -DFE_QSUM_SHFL: the xor butterfly through ds_bpermute.
__device__ __forceinline__ float jsum_dpp(float p);
__device__ __forceinline__ float jsum(float p) {
#ifdef FE_QSUM_SHFL
  p += __shfl_xor(p, 1);
  p += __shfl_xor(p, 2);
  p += __shfl_xor(p, 4);
  p += __shfl_xor(p, 8);
  return p;
#else
  return jsum_dpp(p);
#endif
}
// same sum by DPP row rotations (no LDS crossbar traffic): every lane of the 16-lane row gets the total
__device__ __forceinline__ float jsum_dpp(float p) {
  p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x128, 0xf, 0xf, false));  // row_ror:8
  p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x124, 0xf, 0xf, false));  // row_ror:4
  p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x122, 0xf, 0xf, false));  // row_ror:2
  p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x121, 0xf, 0xf, false));  // row_ror:1
  return p;
}
// Transposing sum over the 16 items of a tile: every lane holds 16 values (index k = 4t + r of a Vec); lane j of each
// 16-lane row returns the total over the row's 16 lanes of value j.  A butterfly that halves the number of values per
// step.  Steps 1, 2: partners 8 and 4 lanes away (row shifts); the half a lane keeps depends on lane bits 3 / 2, i.e. on
// its quad within the row, which is what a DPP bank mask selects (bank = 4 consecutive lanes): two masked
// v_add_f32_dpp per output, no selects.  Steps 3, 4: partners lane^2 and lane^1 (quad permutes), halves selected by lane
// bits 1 / 0.  33 vector instructions instead of the 64 of sixteen jsum_dpp() calls, and the result is spread over the
// lanes: ONE 64-lane LDS atomic adds it to a [64]-feature accumulator (feature of lane (j,q): 16 (j >> 2) + 4 q + (j & 3))
// instead of sixteen 4-lane ones.  Pinned with integer data by fastegnn_selftest_jreduce (tests/test_gpu_toolkit.py).
// (The leading s_nop keeps the two wait states a DPP read needs behind the instruction that produced its operand: the
// hazard recogniser does not look into inline assembly.  EXEC must be all ones.)
__device__ __forceinline__ float jreduce16(const Vec &u) {
  float y[8], z[4], w[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {   // lanes 0-7 of a row keep value k (+ lane+8's), lanes 8-15 value k+8 (+ lane-8's)
    const float a = u.t[k >> 2][k & 3], b = u.t[(k + 8) >> 2][k & 3];
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(y[k]) : "v"(a), "v"(b));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)     // quads 0, 2 keep y[k] (+ lane+4's), quads 1, 3 keep y[k+4] (+ lane-4's)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(z[k]) : "v"(y[k]), "v"(y[k + 4]));
  const int lane = lane_id();
  const bool b1 = lane & 2, b0 = lane & 1;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float keep = b1 ? z[k + 2] : z[k], send = b1 ? z[k] : z[k + 2];
    w[k] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4e, 0xf, 0xf, false));  // quad_perm:[2,3,0,1]
  }
  const float keep = b0 ? w[1] : w[0], send = b0 ? w[0] : w[1];
  return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xb1, 0xf, 0xf, false));  // quad_perm:[1,0,3,2]
}
// row[f] += sum over the 16 items of the tile of u (feature f); `row` is an LDS or global [64] accumulator.  All lanes active.
__device__ __forceinline__ void tile_sum_add(float *row, const Vec &u, int j, int q) {
  const float s = jreduce16(u);   // lane j: the sum over the tile of value j
  atomicAdd(&row[16 * (j >> 2) + 4 * q + (j & 3)], s);
}
// <v, w> over the hidden dimension; every q-lane of the item gets the full dot product
__device__ __forceinline__ float vdot(const Vec &v, const Vec &w) {
  float p = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) p += v.t[t][r] * w.t[t][r];
  return qsum(p);
}

// ---- phase stamps of virt_fwd_kernel (diagnostic builds only: -DFE_STAMP_VF) ----
#ifdef FE_STAMP_VF
__device__ unsigned long long g_vf_stamps[16];
struct VfStamp {
  unsigned long long prev, acc[12];
};
#define VF_TP , VfStamp &_vst
#define VF_TA , _vst
#define VF_T0() VfStamp _vst; for (int _k = 0; _k < 12; ++_k) _vst.acc[_k] = 0; _vst.prev = __builtin_amdgcn_s_memtime();
#define VF_T(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                  _vst.acc[i] += _t - _vst.prev; _vst.prev = _t; __builtin_amdgcn_sched_barrier(0); }
#define VF_TEND() if (lane_id() == 0) { for (int _k = 0; _k < 12; ++_k) atomicAdd(&g_vf_stamps[_k], _vst.acc[_k]); }
#else
#define VF_TP
#define VF_TA
#define VF_T0()
#define VF_T(i)
#define VF_TEND()
#endif
// ---- in-kernel phase stamps (diagnostic builds only: -DFE_STAMP; never in the shipped library) ----
#ifdef FE_STAMP
__device__ unsigned long long g_stamps[16];
struct Stamp {
  unsigned long long prev, acc[12];
};
#define FE_TP , Stamp &_st
#define FE_TA , _st
#define FE_T0() Stamp _st; for (int _k = 0; _k < 12; ++_k) _st.acc[_k] = 0; _st.prev = __builtin_amdgcn_s_memtime();
#define FE_T(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                  _st.acc[i] += _t - _st.prev; _st.prev = _t; __builtin_amdgcn_sched_barrier(0); }
#define FE_TEND() if (lane_id() == 0) { for (int _k = 0; _k < 12; ++_k) atomicAdd(&g_stamps[_k], _st.acc[_k]); }
#elif defined(FE_ISA_MARK)
// assembly-only builds of tools/isa_budget.py: the phase boundaries as comments in the instruction stream (between scheduling
// fences, so that a phase's instructions stay between its marks); never linked into a library
#define FE_TP
#define FE_TA
#define FE_T0()
#define FE_T(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("; FE_MARK " #i); __builtin_amdgcn_sched_barrier(0); }
#define FE_TEND()
#else
#define FE_TP
#define FE_TA
#define FE_T0()
#define FE_T(i)
#define FE_TEND()
#endif

// ---- weight images ----
// Image of A[o][k] (64x64) for gemm64: float index ((t*4+tp)*64 + lane)*4 + r holds
// A[16t + (lane&15)][16tp + 4(lane>>4) + r].
__host__ __device__ __forceinline__ int img_index(int o, int k) {
  int t = o >> 4, i = o & 15, tp = k >> 4, q = (k >> 2) & 3, r = k & 3;
  return ((t * 4 + tp) * 64 + q * 16 + i) * 4 + r;
}

// acc[o][item] += sum_k A[o][k] * in[k][item].  img may live in LDS or global memory.
__device__ __forceinline__ void gemm64(const float *img, const Vec &in, Vec &acc) {
  const f32x4 *ip = reinterpret_cast<const f32x4 *>(img) + lane_id();
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) {
    f32x4 a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = ip[(t * 4 + tp) * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][r], in.t[tp][r], acc.t[t], 0, 0, 0);
  }
}

// gemm64 between scheduling fences: the operand reads of ONE layer are in flight at a time.  Without the
// fences the scheduler hoists the image reads of every following layer of a tile (64 registers each).
__device__ __forceinline__ void gemm64_f(const float *img, const Vec &in, Vec &acc) {
  __builtin_amdgcn_sched_barrier(0);
  gemm64(img, in, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- 3-way bf16 split ("bf16x3") variant of gemm64 ------------------------------------------
// fp32-input MFMA shares the vector ALUs with every other VALU instruction and runs at 1/16 of the
// bf16 matrix rate.  x = h + m + l with h, m, l bf16 values (8+8+8 mantissa bits, each the round-to-nearest
// image of the residual so far -- part_pack below; the residuals are exact, so the three parts reproduce x to 2^-25) turns one fp32 product into six bf16
// products (hh, hm, mh, hl, lh, mm; the dropped ml, lm, ll terms are <= 2^-24 relative, i.e. below
// fp32 rounding) accumulated in fp32 by v_mfma_f32_16x16x32_bf16 on the matrix pipe, which runs
// beside the VALU.  Same D layout as gemm64, so the chained-MLP property is unchanged.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int IMG3 = 3 * 2048;   // u32 words of a split image: 3 parts x (64x64 bf16)

struct Split {
  u32x4 p[3][2];   // [part h|m|l][k-step] : 8 bf16 per lane
};
__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float trunc_bf(float x) { return __builtin_bit_cast(float, f2u(x) & 0xffff0000u); }
// {bf16(x0) in the low half, bf16(x1) in the high half}, by truncation
__device__ __forceinline__ unsigned pack_hi(float x0, float x1) {
  return __builtin_amdgcn_perm(f2u(x1), f2u(x0), 0x07060302u);
}
// One level of the h | m | l split of a pair: returns the packed bf16 parts of (a0, a1) and leaves the residuals in them.
// Round to nearest even: one v_cvt_pk_bf16_f32 per pair, shift / and + sub per element -- the instruction count of the
// truncating split of rounds 1-2 (and + sub per element, one perm per pair; -DFE_SPLIT_TRUNC brings it back).  With
// truncation every part has the sign of the value, so the dropped m*l, l*m, l*l terms and the last residue all point
// towards zero: a bias of ~1e-7 per product that does not cancel in cancelling sums.  Measured on one box (round 3,
// tools/gpu_lever_newton.sh): 40 -> 32 gradient comparisons beyond 2 x ref + 1e-6, the attention goldens' excesses down by
// 2-4 x (att_mlp.0.bias 2.67e-5 -> 6.8e-6), for +0.5 % step time (v_cvt_pk_bf16_f32 issues in 4-5 cycles, v_perm_b32 in 4).
__device__ __forceinline__ unsigned part_pack(float &a0, float &a1) {
#ifndef FE_SPLIT_TRUNC
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{a0, a1}, bf16x2_));
  a0 -= __builtin_bit_cast(float, p << 16);
  a1 -= __builtin_bit_cast(float, p & 0xffff0000u);
  return p;
#else
  const unsigned p = pack_hi(a0, a1);
  a0 -= trunc_bf(a0);
  a1 -= trunc_bf(a1);
  return p;
#endif
}
// k index held by element e of lane quarter q in k-step s (matches the chained D layout)
__host__ __device__ __forceinline__ int bf3_k(int s, int q, int e) { return 16 * (2 * s + (e >> 2)) + 4 * q + (e & 3); }
// u32 index of (part, out o, k) in a split image; pairs (e, e+1) share a word
__host__ __device__ __forceinline__ int img3_index(int part, int o, int k) {
  const int t = o >> 4, i = o & 15, tile = k >> 4, q = (k >> 2) & 3, r = k & 3;
  const int s = tile >> 1, e = ((tile & 1) << 2) | r;
  return part * 2048 + ((t * 2 + s) * 64 + q * 16 + i) * 4 + (e >> 1);
}
// word (two bf16) of the split image: elements k0 (low half) and k0+1 (high half) of part `part`
__device__ __forceinline__ unsigned split_word(float w0, float w1, int part) {
  float a0 = w0, a1 = w1;
  unsigned p = 0;
  for (int k = 0; k <= part; ++k) p = part_pack(a0, a1);
  return p;
}
__device__ __forceinline__ Split vsplit(const Vec &v) {
  Split S;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = v.t[2 * s + (e >> 2)][e & 3];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      S.p[0][s][w] = part_pack(x[2 * w], x[2 * w + 1]);
      S.p[1][s][w] = part_pack(x[2 * w], x[2 * w + 1]);
      S.p[2][s][w] = part_pack(x[2 * w], x[2 * w + 1]);
    }
  }
  return S;
}
// img3: split image, IMG3 words = parts h | m | l (2048 words each), normally LDS resident.
// (fragments of group g+1 requested ahead of the MFMAs of group g: see gemm64_x3_rm)
__device__ __forceinline__ void gemm64_x3(const unsigned *img3, const Split &in, Vec &acc) {
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(img3) + lane_id();
#ifdef FE_NO_GEMM_PIPE
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ip[(t * 2 + s) * 64]);
      const bf16x8 am = __builtin_bit_cast(bf16x8, ip[512 + (t * 2 + s) * 64]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, ip[1024 + (t * 2 + s) * 64]);
      const bf16x8 xh = __builtin_bit_cast(bf16x8, in.p[0][s]);
      const bf16x8 xm = __builtin_bit_cast(bf16x8, in.p[1][s]);
      const bf16x8 xl = __builtin_bit_cast(bf16x8, in.p[2][s]);
      // smallest terms first: (weight part, activation part) = (l,h) (m,m) (h,l) | (m,h) (h,m) | (h,h)
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xm, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xh, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xm, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc.t[t], 0, 0, 0);
    }
#else
  __builtin_amdgcn_sched_barrier(0);
  u32x4 fr[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) fr[0][p] = ip[p * 512];
  __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // 3 DS reads (one ds_read_b128 per fragment)
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int s = g >> 2, t = g & 3, cb = g & 1;
    if (g + 1 < 8) {
      const int sn = (g + 1) >> 2, tn = (g + 1) & 3;
#pragma unroll
      for (int p = 0; p < 3; ++p) fr[cb ^ 1][p] = ip[p * 512 + (tn * 2 + sn) * 64];
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
    const bf16x8 ah = __builtin_bit_cast(bf16x8, fr[cb][0]), am = __builtin_bit_cast(bf16x8, fr[cb][1]),
                 al = __builtin_bit_cast(bf16x8, fr[cb][2]);
    const bf16x8 xh = __builtin_bit_cast(bf16x8, in.p[0][s]);
    const bf16x8 xm = __builtin_bit_cast(bf16x8, in.p[1][s]);
    const bf16x8 xl = __builtin_bit_cast(bf16x8, in.p[2][s]);
    // smallest terms first: (weight part, activation part) = (l,h) (m,m) (h,l) | (m,h) (h,m) | (h,h)
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xm, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xh, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xm, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc.t[t], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // 6 MFMAs
  }
  __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ void gemm64_x3(const unsigned *img3, const Vec &in, Vec &acc) { gemm64_x3(img3, vsplit(in), acc); }

// Diagnostic lever (-DFE_EDGE_T2, edge_fwd only): the activation operand split into TWO bf16 parts
// (h, m: 16 mantissa bits) -- 3.5 instead of 5.5 vector instructions per element and five products instead of six (the
// (h, l) product is gone).  Costs 2^-17 relative on the operand; measured in DESIGN.md section 12, not used by any product path.
__device__ __forceinline__ void gemm64_x3_t2(const unsigned *img3, const Vec &v, Vec &acc) {
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(img3) + lane_id();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
    u32x4 ph, pm;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = v.t[2 * s + (e >> 2)][e & 3];
#pragma unroll
    for (int w = 0; w < 4; ++w) {   // (part_pack: the rounding mode of the three-part split, nearest since late round 3)
      ph[w] = part_pack(x[2 * w], x[2 * w + 1]);
      pm[w] = part_pack(x[2 * w], x[2 * w + 1]);
    }
    const bf16x8 xh = __builtin_bit_cast(bf16x8, ph), xm = __builtin_bit_cast(bf16x8, pm);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ip[(t * 2 + s) * 64]);
      const bf16x8 am = __builtin_bit_cast(bf16x8, ip[512 + (t * 2 + s) * 64]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, ip[1024 + (t * 2 + s) * 64]);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xm, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xh, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xm, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc.t[t], 0, 0, 0);
    }
  }
}

// ---- bf16 operand mode (FASTEGNN_F_BF16): both operands rounded to bf16 (RNE), ONE bf16 product, fp32 accumulate ----
// The weights are rounded by pack_kernel, so the `h` part of a split image IS the bf16 weight matrix (m = l = 0) and
// the fp32 images hold bf16-representable values; activations are rounded here (v_cvt_pk_bf16_f32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_rne(float x0, float x1) {   // {bf16(x0) low half, bf16(x1) high half}
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
}
__device__ __forceinline__ float round_bf(float x) { return (float)(__bf16)x; }
__host__ __device__ __forceinline__ float round_bf_host(float x) {   // same rounding by integer arithmetic (finite x)
  unsigned u = __builtin_bit_cast(unsigned, x);
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  return __builtin_bit_cast(float, u);
}
struct BfOp {
  u32x4 p[2];   // [k-step] : 8 bf16 per lane, the k order of the split images (bf3_k)
};
__device__ __forceinline__ BfOp vpack_bf(const Vec &v) {
  BfOp B;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e = 2 * w;
      B.p[s][w] = pack_rne(v.t[2 * s + (e >> 2)][e & 3], v.t[2 * s + ((e + 1) >> 2)][(e + 1) & 3]);
    }
  return B;
}
__device__ __forceinline__ Vec vround(const Vec &v) { return vmap(v, [](float z) { return round_bf(z); }); }
// img3: split image whose h part (first 2048 words) holds the bf16 weights
__device__ __forceinline__ void gemm64_b1(const unsigned *img3, const BfOp &in, Vec &acc) {
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(img3) + lane_id();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, in.p[s]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ip[(t * 2 + s) * 64]), x, acc.t[t], 0, 0, 0);
  }
}

// ---- row-major split images: ONE LDS copy of a weight serves the product AND its transpose ------------------------
// Part p (h | m | l) of W [64 out x 64 in] as plain rows of bf16: element (p, o, k) at byte p*RM_PART + o*RM_RS + 2k.
// The product y = W x reads a lane's A fragment -- 8 bf16 of row o = 16t + i, k = 32s + 4q + {0..3} and + 16 -- with two
// ds_read_b64; the transposed product y = W^T g reads it with two ds_read_b64_tr_b16 (gfx950: a 16-lane group fetches a
// block of 4 rows x 16 columns and every lane receives one COLUMN of it): rows o = 32s + 4q + {0..3} (and + 16) at column
// k = 16t + i.  Both orders equal bf3_k, the k order of the chained D layout, so either product takes the usual operand.
// The 144-byte row stride keeps the row reads conflict-free (bank = 36 o + 2 q mod 64 over a 32-lane half) and leaves the
// transposed reads 2-way on 4 of 64 banks.
constexpr int RM_RS = 144;
constexpr int RM_PART = 64 * RM_RS;       // 9 216 bytes
constexpr int RM_BYTES = 3 * RM_PART;     // 27 648 bytes = 27 wave-instructions of an LDS-DMA copy
constexpr int RM_WORDS = RM_BYTES / 4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 lds_tr_read(const char *p) {   // EXEC must be all ones (the gather crosses lanes)
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                       (__attribute__((address_space(3))) s16x4 *)(p)));
}
// A fragment of part `part`, tile t, k-step s of W (TR = false) or W^T (TR = true)
template <bool TR>
__device__ __forceinline__ bf16x8 rm_frag(const char *img, int part, int t, int s) {
  const int l = lane_id(), i = l & 15, q = l >> 4;
  u32x2 lo, hi;
  if constexpr (TR) {
    const char *p = img + part * RM_PART + (32 * s + 4 * q + (i >> 2)) * RM_RS + (16 * t + 4 * (i & 3)) * 2;
    lo = lds_tr_read(p);
    hi = lds_tr_read(p + 16 * RM_RS);
  } else {
    const char *p = img + part * RM_PART + (16 * t + i) * RM_RS + (32 * s + 4 * q) * 2;
#ifdef FE_RM_READ2
    lo = *reinterpret_cast<const u32x2 *>(p);
    hi = *reinterpret_cast<const u32x2 *>(p + 32);
#else
    // Two ds_read_b64 (conflict-free on the 144-byte rows, 2 LDS cycles each).  As plain loads the compiler fuses them --
    // and the loads of neighbouring fragments -- into ds_read2_b64, which is banked modulo 32 dwords: rows i and i + 8 of
    // a 16-lane group then collide (2-way) and the instruction runs at half the rate; the counters showed a third of the
    // backward kernels' LDS cycles as bank conflicts.  Volatile accesses are not merged (and stay under the compiler's
    // own s_waitcnt bookkeeping, unlike inline assembly).
    typedef __attribute__((address_space(3))) const volatile u32x2 *lds_vu32x2;
    lo = *(lds_vu32x2)(p);
    hi = *(lds_vu32x2)(p + 32);
#endif
  }
  return __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
}
// The eight (k-step, out-tile) groups of a product are software-pipelined by hand: the three fragments of group g+1 are
// requested BEFORE the six MFMAs of group g are issued (sched_group_barrier pins the order; the compiler's own schedule was
// "6 MFMA, 6 reads, wait" -- every group then waited a full LDS round trip with an idle matrix pipe behind it).  Costs 12
// more live registers: PIPE = false (edge_bwd_pc_kernel, whose producers are at their register limit: 10 spilled registers
// and 3.29 instead of 3.21 ms per step with it) or -DFE_NO_GEMM_PIPE restore the compiler's order.  Measured on one box
// (tools/gpu_ab.sh): virt_bwd 4.15 -> 3.94 ms per step, virt_fwd 1.81 -> 1.77.
template <bool TR, bool PIPE = true>
__device__ __forceinline__ void gemm64_x3_rm(const char *img, const Split &in, Vec &acc) {
#ifdef FE_NO_GEMM_PIPE
  constexpr bool pipe = false;
#else
  constexpr bool pipe = PIPE;
#endif
  if constexpr (!pipe) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 ah = rm_frag<TR>(img, 0, t, s), am = rm_frag<TR>(img, 1, t, s), al = rm_frag<TR>(img, 2, t, s);
      const bf16x8 xh = __builtin_bit_cast(bf16x8, in.p[0][s]);
      const bf16x8 xm = __builtin_bit_cast(bf16x8, in.p[1][s]);
      const bf16x8 xl = __builtin_bit_cast(bf16x8, in.p[2][s]);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xm, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xh, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xm, acc.t[t], 0, 0, 0);
      acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc.t[t], 0, 0, 0);
    }
  } else {
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 fr[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) fr[0][p] = rm_frag<TR>(img, p, 0, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // 6 DS reads (two ds_read_b64 per fragment)
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int s = g >> 2, t = g & 3, cb = g & 1;
    if (g + 1 < 8) {
#pragma unroll
      for (int p = 0; p < 3; ++p) fr[cb ^ 1][p] = rm_frag<TR>(img, p, (g + 1) & 3, (g + 1) >> 2);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
    }
    const bf16x8 ah = fr[cb][0], am = fr[cb][1], al = fr[cb][2];
    const bf16x8 xh = __builtin_bit_cast(bf16x8, in.p[0][s]);
    const bf16x8 xm = __builtin_bit_cast(bf16x8, in.p[1][s]);
    const bf16x8 xl = __builtin_bit_cast(bf16x8, in.p[2][s]);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xm, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xh, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xm, acc.t[t], 0, 0, 0);
    acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc.t[t], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // 6 MFMAs
  }
  __builtin_amdgcn_sched_barrier(0);
  }
}
template <bool TR>
__device__ __forceinline__ void gemm64_b1_rm(const char *img, const BfOp &in, Vec &acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, in.p[s]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc.t[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rm_frag<TR>(img, 0, t, s), x, acc.t[t], 0, 0, 0);
  }
}

// ---- one k-block (8 values per lane) of a weight-gradient contraction, as bf16 parts ----
struct Split8 {
  u32x4 h, m, l;
};
// eight fp32 values -> their three bf16 parts, packed as one k-block of v_mfma_f32_16x16x32_bf16
__device__ __forceinline__ Split8 split8(const float (&x_)[8]) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = x_[e];
  Split8 S;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    S.h[w] = part_pack(x[2 * w], x[2 * w + 1]);
    S.m[w] = part_pack(x[2 * w], x[2 * w + 1]);
    S.l[w] = part_pack(x[2 * w], x[2 * w + 1]);
  }
  return S;
}
// eight fp32 values -> one k-block of bf16 values, round to nearest even (bf16 operand mode)
__device__ __forceinline__ u32x4 round8(const float (&x)[8]) {
  u32x4 r;
#pragma unroll
  for (int w = 0; w < 4; ++w) r[w] = pack_rne(x[2 * w], x[2 * w + 1]);
  return r;
}

// ---- 2-way fp16 split ("f16x2") of the FORWARD kernels' products (-DFE_FWD_F16=mask) -------------
// x = h + l / 2^11 with h = fp16(x) (RNE) and l = fp16((x - h) * 2^11): the residual is exact in fp32, 11 + 11 significant
// bits and a signed residual leave |x - h - l / 2^11| <= 2^-23 of x's binade (one bit short of fp32) down to |x| = 2^-14;
// below that the absolute error is <= 2^-36.  One fp32 product becomes THREE fp16 products -- (l_w, h_x) and (h_w, l_x) into
// an accumulator that carries the 2^11, folded into the result by one fma per output element, then (h_w, h_x); the dropped
// (l, l) term is <= 2^-24 of |w||x| -- where bf16x3 needs six, and the operand split is 2.5 vector instructions per element
// (v_cvt_pk_f16_f32, v_fma_mix_f32 for the residual straight from the packed half, v_fma_mixlo/hi_f16 for the scaled low
// part) where the three-part bf16 split needs 5.5.  fp16 carries 5 exponent bits: operands must stay below 65 504 --
// activations and weights do, gradients (1e-9 and below on the headline frame) do not: the backward kernels split them
// relative to a per-item power of two (Split2s / vsplit2_scaled below).  Images: the img3 layout with part 0 = h and
// part 1 = l (part 2 unused).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F2_UP = 2048.f, F2_DOWN = 1.0f / 2048.f;
struct Split2 {
  u32x4 p[2][2];   // [part h | l][k-step] : 8 fp16 per lane
};
// packed fp16 parts of the pair (a0 low half, a1 high half)
__device__ __forceinline__ void part2_pack(float a0, float a1, unsigned &ph, unsigned &pl) {
  const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, f16x2));
  float r0, r1;
  // r = a - float(h): src0 is read as the low / high half of the packed pair (op_sel_hi: f16 source, op_sel: which half)
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(a0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(a1));
  unsigned l;
  // (ends with the VALU -> MFMA-operand wait states: the weight images go to memory, but edge_fwd32.hip feeds these to MFMAs; see vsplit2)
  asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0\n\ts_nop 1" : "=&v"(l) : "v"(r0), "v"(r1), "s"(F2_UP));
  ph = h;
  pl = l;
}
// word (two fp16) of an f16x2 image: elements k0 (low half), k0 + 1 (high half) of part 0 (h) or 1 (l)
__device__ __forceinline__ unsigned split2_word(float w0, float w1, int part) {
  unsigned h, l;
  part2_pack(w0, w1, h, l);
  return part == 0 ? h : l;
}
// Inline assembly is opaque to the compiler's hazard recognizer: a register written inside an asm string and read by an MFMA as its
// A / B operand needs the two VALU -> MFMA-operand wait states INSIDE the string (cdna_hip_programming.md section 5.7 item 2).  The
// scaled low parts of a k-step (and the scaled high parts of vsplit2_scaled / wg32_split) are therefore written by ONE asm block per
// four words that ends with `s_nop 1`.  Rounds 4-5 had one block per word and no pad: the default machine scheduler happened to keep
// two instructions between the last block and the first MFMA; -amdgpu-sched-strategy=max-memory-clause did not (stale low parts:
// errors of 2^-11 relative, the EGNN golden and smoke() at 1e-4 -- found in round 5, tools/gpu_r5_sched_bisect.sh).
// Round 6: tools/isa_hazards.py checks the DISASSEMBLY of every built library for this (and the other hand-offs asm can break);
// -DFE_HAZARD_SELFTEST drops the pad so that tests/test_isa_hazards_cpu.py can see the checker fail on a tree without it.
#ifdef FE_HAZARD_SELFTEST
#define FE_MIX_PAD ""
#else
#define FE_MIX_PAD "s_nop 1"
#endif
// l[w] = {fp16(r[2w] * sc), fp16(r[2w+1] * sc)} for w = 0..3
// IN_PLACE (the residuals are dead afterwards -- the low-part blocks): l[w] is tied to the register of r[2w].  Round 6: the binary hazard
// checker found an "=&v" output of this block allocated to a register that a 4-pass MFMA had written two instructions earlier (the low-part
// accumulator of the previous product, dead once the next MFMA had taken it as C): v_fma_mixlo_f16 reads and rewrites its destination,
// and an MFMA's write-back needs its wait states before ANY later access (the compiler pads its own instructions, not an asm string).
// Tied to its input the output lands in a register that the vector ALU wrote last.
template <bool IN_PLACE = false>
__device__ __forceinline__ void mix_pack4(const float (&r)[8], float sc, unsigned (&l)[4]) {
  if constexpr (IN_PLACE) {
#pragma unroll
    for (int w = 0; w < 4; ++w) l[w] = f2u(r[2 * w]);
    asm("v_fma_mixlo_f16 %0, %0, %8, 0\n\tv_fma_mixhi_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %1, %8, 0\n\tv_fma_mixhi_f16 %1, %5, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %2, %8, 0\n\tv_fma_mixhi_f16 %2, %6, %8, 0\n\t"
        "v_fma_mixlo_f16 %3, %3, %8, 0\n\tv_fma_mixhi_f16 %3, %7, %8, 0\n\t"
        FE_MIX_PAD
        : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3])
        : "v"(r[1]), "v"(r[3]), "v"(r[5]), "v"(r[7]), "v"(sc));
  } else {
  asm("v_fma_mixlo_f16 %0, %4, %12, 0\n\tv_fma_mixhi_f16 %0, %5, %12, 0\n\t"
      "v_fma_mixlo_f16 %1, %6, %12, 0\n\tv_fma_mixhi_f16 %1, %7, %12, 0\n\t"
      "v_fma_mixlo_f16 %2, %8, %12, 0\n\tv_fma_mixhi_f16 %2, %9, %12, 0\n\t"
      "v_fma_mixlo_f16 %3, %10, %12, 0\n\tv_fma_mixhi_f16 %3, %11, %12, 0\n\t"
      FE_MIX_PAD
      : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
      : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(sc));
  }
}
__device__ __forceinline__ Split2 vsplit2(const Vec &v) {
  Split2 S;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float r[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e = 2 * w;
      const float a0 = v.t[2 * s + (e >> 2)][e & 3], a1 = v.t[2 * s + ((e + 1) >> 2)][(e + 1) & 3];
      const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, f16x2));   // (compiler-visible: padded by hipcc)
      // r = a - float(h): src0 is read as the low / high half of the packed pair (op_sel_hi: f16 source, op_sel: which half)
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[e]) : "v"(h), "v"(a0));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[e + 1]) : "v"(h), "v"(a1));
      S.p[0][s][w] = h;
    }
    unsigned l[4];
    mix_pack4<true>(r, F2_UP, l);
#pragma unroll
    for (int w = 0; w < 4; ++w) S.p[1][s][w] = l[w];
  }
  return S;
}
// acc += W x on an f16x2 image (img3 layout, parts h | l).  Output tile by output tile: the two cross products of both
// k-steps into a 4-register accumulator, one fma per element folds it (scaled by 2^-11) into acc, then the (h, h) products.
// PIPE: the software-pipelined order of gemm64_f2_rm_ below -- next tile's fragments requested ahead, independent low / high chains,
// fold one tile late; see there.  Measured on one box (profiles/r06_lever_f2_pipe.txt): edge_fwd 0.802 -> 0.793 ms per step with it,
// virt_fwd 1.246 -> 1.268 without it: the edge kernel takes it (-DFE_F2_PIPE_FWD=0 switches it off), the virtual kernel does not.
#ifndef FE_F2_PIPE_FWD
#define FE_F2_PIPE_FWD 1
#endif
// off4: a run-time offset of the image in 16-byte units, added to the LANE index rather than to the pointer (virt_fwd's PAIR walk picks one
// of four stage slots per iteration: as pointer arithmetic ahead of the generic cast that crashed this hipcc's backend, layer_fwd.hip)
template <bool PIPE = false>
__device__ __forceinline__ void gemm64_f2(const unsigned *img3, const Split2 &in, Vec &acc, int off4 = 0) {
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(img3) + (lane_id() + off4);
  const f16x8 xh0 = __builtin_bit_cast(f16x8, in.p[0][0]), xh1 = __builtin_bit_cast(f16x8, in.p[0][1]);
  const f16x8 xl0 = __builtin_bit_cast(f16x8, in.p[1][0]), xl1 = __builtin_bit_cast(f16x8, in.p[1][1]);
  if constexpr (PIPE && FE_F2_PIPE_FWD != 0) {
  __builtin_amdgcn_sched_barrier(0);
  u32x4 fr[2][4];   // [buffer][ah0 | ah1 | al0 | al1]
#pragma unroll
  for (int k = 0; k < 4; ++k) fr[0][k] = ip[(k >> 1) * 512 + (k & 1) * 64];
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 DS reads (one ds_read_b128 per fragment)
  f32x4 lo_p = {0.f, 0.f, 0.f, 0.f}, hi_p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int cb = t & 1;
    if (t + 1 < 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fr[cb ^ 1][k] = ip[(k >> 1) * 512 + ((t + 1) * 2 + (k & 1)) * 64];
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
    const f16x8 ah0 = __builtin_bit_cast(f16x8, fr[cb][0]), ah1 = __builtin_bit_cast(f16x8, fr[cb][1]);
    const f16x8 al0 = __builtin_bit_cast(f16x8, fr[cb][2]), al1 = __builtin_bit_cast(f16x8, fr[cb][3]);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f};
    f32x4 hi = acc.t[t];
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xh0, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh0, hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xl0, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh1, hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xh1, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xl1, lo, 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // 6 MFMAs
    if (t > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc.t[t - 1][r] = __builtin_fmaf(lo_p[r], F2_DOWN, hi_p[r]);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU
    }
    lo_p = lo;
    hi_p = hi;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc.t[3][r] = __builtin_fmaf(lo_p[r], F2_DOWN, hi_p[r]);
  __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const f16x8 ah0 = __builtin_bit_cast(f16x8, ip[(t * 2 + 0) * 64]), ah1 = __builtin_bit_cast(f16x8, ip[(t * 2 + 1) * 64]);
    const f16x8 al0 = __builtin_bit_cast(f16x8, ip[512 + (t * 2 + 0) * 64]), al1 = __builtin_bit_cast(f16x8, ip[512 + (t * 2 + 1) * 64]);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xh0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xl0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xh1, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xl1, lo, 0, 0, 0);
    f32x4 hi = acc.t[t];
#pragma unroll
    for (int r = 0; r < 4; ++r) hi[r] = __builtin_fmaf(lo[r], F2_DOWN, hi[r]);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh0, hi, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh1, hi, 0, 0, 0);
    acc.t[t] = hi;
  }
  }
}

// -DFE_LOG2E_FOLD=mask (bit 0: edge_fwd): log2(e) folded into the packed weights.  silu(z) = z / (1 + exp2(-z log2 e)) spends one
// of its five vector instructions on the argument scale; a pre-activation that comes out of an in-kernel product can arrive
// in units of ln 2 instead (z2 = z log2 e: the image of that product and its bias are pre-scaled), silu2(z2) = z2 / (1 + exp2(-z2))
// = silu(z) log2 e needs no scale, and the factor rides along: the NEXT product takes the scaled activation with unscaled
// weights and a scaled bias, head vectors and the aggregation's 1/deg absorb 1 / log2 e.  The scaled image is split from the
// double-precision product w * log2(e), so no weight is rounded twice.
// Bit 1 (round 5): the FIRST layer as well -- node_pre_fwd stores P and Q in units of ln 2 (its W1a / W1b images and b1 are pre-scaled,
// pack.hip), the edge kernels scale the radial / edge_attr features of the rank-3 update by log2(e), so the first SiLU of a tile takes
// its argument as it arrives (16 vector multiplies per 16-edge tile less: profiles/r05_edge_fwd_instruction_budget.txt).  The factor
// rides through the first product: W2's image is then UNSCALED (its input is already log2(e) too large).  The backward kernels read the
// same P / Q and multiply by ln 2 where the forward multiplied by log2(e) (silu_both2): no instruction more.  Gradients are taken
// with respect to the TRUE P / Q (the scaled storage is a representation), so node_pre_bwd and the weight gradients are untouched.
#ifndef FE_LOG2E_FOLD
#define FE_LOG2E_FOLD 3   // bit 0 adopted in round 4 (edge_fwd 0.870 -> 0.854 ms per step; profiles/r04_lever_f16x2_fwd_bwd_fold.txt), bit 1 in round 5
#endif
// (the generic-activation build keeps the plain form: the fold is a property of SiLU)
#ifdef FE_ACT_GENERIC
constexpr bool LOG2E_FOLD_EDGE = false;
constexpr bool LOG2E_FOLD_FIRST = false;
#else
constexpr bool LOG2E_FOLD_EDGE = (FE_LOG2E_FOLD & 1) != 0;
constexpr bool LOG2E_FOLD_FIRST = (FE_LOG2E_FOLD & 2) != 0;
static_assert(!LOG2E_FOLD_FIRST || LOG2E_FOLD_EDGE, "FE_LOG2E_FOLD: bit 1 (first layer) needs bit 0");
#endif
constexpr float LOG2E_F = 1.4426950408889634f, LN2_F = 0.6931471805599453f;
constexpr double LOG2E_D = 1.4426950408889634074;
__device__ __forceinline__ float silu2_f(float z2) { return z2 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z2)); }
__device__ __forceinline__ Vec vsilu2(const Vec &a) { return vmap(a, [](float z) { return silu2_f(z); }); }
// y = silu(z), z <- silu'(z) from a pre-activation given in units of ln 2 (z2 = z log2 e): the backward's reading of a folded P + Q
__device__ __forceinline__ Vec vsilu_keep_d2(Vec &z2) {
  Vec y;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z2.t[t][r]));
      const float yy = (z2.t[t][r] * LN2_F) * s;
      y.t[t][r] = yy;
      z2.t[t][r] = s + yy * (1.0f - s);
    }
  return y;
}
// word of a split image from two DOUBLE values (a scaled weight): parts as part_pack / part2_pack make them, the residuals
// taken in double
__device__ __forceinline__ unsigned split_word_d(double w0, double w1, int part, bool f16) {
  unsigned out = 0;
  double r[2] = {w0, w1};
  if (f16) {
    for (int e = 0; e < 2; ++e) {
      const _Float16 h = (_Float16)(float)r[e];
      const _Float16 l = (_Float16)(float)((r[e] - (double)(float)h) * (double)F2_UP);
      const unsigned short bits = __builtin_bit_cast(unsigned short, part == 0 ? h : l);
      out |= (unsigned)bits << (16 * e);
    }
    return part < 2 ? out : 0u;
  }
  for (int e = 0; e < 2; ++e) {
    unsigned short bits = 0;
    for (int k = 0; k <= part; ++k) {
      const __bf16 b = (__bf16)(float)r[e];
      bits = __builtin_bit_cast(unsigned short, b);
      r[e] -= (double)(float)b;
    }
    out |= (unsigned)bits << (16 * e);
  }
  return out;
}

// ---- f16x2 for GRADIENT operands: per-item power-of-two scale ---------------------------------------------------------------
// Gradients sit far below fp16's range (1e-9 on the headline frame): the 64 values of an item are multiplied by s = 2^k, k chosen
// from the item's largest magnitude so that it lands in [2^14, 2^15), split as above, and the product is multiplied by 1/s (both
// exact).  Errors are then 2^-23 of the item's LARGEST component, which is what an fp32 dot product over the item gives too.
struct Split2s {
  Split2 s;
  float inv;   // 1 / scale of this lane's item
};
__device__ __forceinline__ float qmax(float p) {   // max over the four q-lanes of an item (as qsum)
  float a = p, b = p;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  p = fmaxf(a, b);
  a = p; b = p;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ Split2s vsplit2_scaled(const Vec &v) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v.t[t][r]));
  m = qmax(m);
  // m in [2^(e-127), 2^(e-126)): scale 2^(141-e) puts it into [2^14, 2^15).  Exponent fields clamped to the normal range: items
  // whose largest component is below 2^-113 (or zero) get inv = 0 -- their product is dropped, it is below fp32's range anyway.
  int se = 268 - (int)(f2u(m) >> 23);
  se = se > 254 ? 254 : (se < 1 ? 1 : se);
  const float sc = __builtin_bit_cast(float, (unsigned)se << 23);
  Split2s S;
  S.inv = __builtin_bit_cast(float, (unsigned)(254 - se) << 23);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float a[8], r[8];
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = v.t[2 * s + (e >> 2)][e & 3];
    mix_pack4(a, sc, h);   // (h and l are MFMA operands written by inline assembly: blocks that end with the wait states, see vsplit2)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r[2 * w]) : "v"(a[2 * w]), "v"(sc), "v"(h[w]));
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[2 * w + 1]) : "v"(a[2 * w + 1]), "v"(sc), "v"(h[w]));
    }
    mix_pack4<true>(r, F2_UP, l);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      S.s.p[0][s][w] = h[w];
      S.s.p[1][s][w] = l[w];
    }
  }
  return S;
}
// f16x2 products on a ROW-MAJOR image whose parts 0 | 1 hold the fp16 h | l of the weight (pack.hip, slots RM_F16 + k): plain and
// transposed reads as for the bf16 parts (16-bit elements either way).  SCALED: the operand is a Split2s, the result is
// multiplied by the item's 1 / scale before it is added to acc.
// Round 6: the four output tiles of a product are software-pipelined by hand (FE_F2_PIPE, default on).  The compiler's own schedule
// reused ONE set of fragment registers for every tile -- read 8 fragments, wait out the LDS round trip, four dependent MFMAs on `lo`,
// s_nop 7, fold, two dependent MFMAs on `hi`, then the next tile's reads: four exposed LDS latencies and four exposed MFMA -> VALU
// drains per product, with only two waves per SIMD to cover them (the backward producers ran at 0.55 of the issue slots, each wave at
// its solo latency-bound speed).  Here the fragments of tile t + 1 are requested BEFORE the MFMAs of tile t (double buffer, +16
// registers), the low-part chain and the (h, h) chain of a tile are independent accumulators issued interleaved, and the fold of tile t
// (hi + lo / 2^11) is issued behind the MFMAs of tile t + 1, when its operands have long left the matrix pipe.  The (h, h) chain now
// starts from acc and the fold adds the low part last: the same sum in another order.  -DFE_F2_PIPE=0 restores the old form.
// Measured on one box each, ms per step at cfg4 (profiles/r06_lever_f2_pipe.txt): edge_bwd 2.935 -> 2.906, virt_bwd (phased form) 3.753 -> 3.700;
// the tile-major virt_bwd_pc, already at 256 registers with 12 spilled, LOSES with it (24 spilled: 2.93 -> 3.01) and passes P16 = false.
#ifndef FE_F2_PIPE
#define FE_F2_PIPE 1
#endif
template <bool TR, bool SCALED, bool PIPE = true>
__device__ __forceinline__ void gemm64_f2_rm_(const char *img, const Split2 &in, float inv, Vec &acc) {
  const f16x8 xh0 = __builtin_bit_cast(f16x8, in.p[0][0]), xh1 = __builtin_bit_cast(f16x8, in.p[0][1]);
  const f16x8 xl0 = __builtin_bit_cast(f16x8, in.p[1][0]), xl1 = __builtin_bit_cast(f16x8, in.p[1][1]);
  if constexpr (PIPE && FE_F2_PIPE != 0) {
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 fr[2][4];   // [buffer][ah0 | ah1 | al0 | al1]
#pragma unroll
  for (int k = 0; k < 4; ++k) fr[0][k] = rm_frag<TR>(img, k >> 1, 0, k & 1);
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // 8 DS reads (two ds_read_b64 per fragment)
  f32x4 lo_p = {0.f, 0.f, 0.f, 0.f}, hi_p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int cb = t & 1;
    if (t + 1 < 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fr[cb ^ 1][k] = rm_frag<TR>(img, k >> 1, t + 1, k & 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    }
    const f16x8 ah0 = __builtin_bit_cast(f16x8, fr[cb][0]), ah1 = __builtin_bit_cast(f16x8, fr[cb][1]);
    const f16x8 al0 = __builtin_bit_cast(f16x8, fr[cb][2]), al1 = __builtin_bit_cast(f16x8, fr[cb][3]);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f};
    f32x4 hi = SCALED ? f32x4{0.f, 0.f, 0.f, 0.f} : acc.t[t];
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xh0, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh0, hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xl0, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh1, hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xh1, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xl1, lo, 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // 6 MFMAs
    if (t > 0) {   // the fold of the PREVIOUS tile: its accumulators left the matrix pipe while this tile's MFMAs were issued
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (SCALED) acc.t[t - 1][r] = __builtin_fmaf(__builtin_fmaf(lo_p[r], F2_DOWN, hi_p[r]), inv, acc.t[t - 1][r]);
        else acc.t[t - 1][r] = __builtin_fmaf(lo_p[r], F2_DOWN, hi_p[r]);
      }
      __builtin_amdgcn_sched_group_barrier(0x002, SCALED ? 8 : 4, 0);   // VALU
    }
    lo_p = lo;
    hi_p = hi;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if constexpr (SCALED) acc.t[3][r] = __builtin_fmaf(__builtin_fmaf(lo_p[r], F2_DOWN, hi_p[r]), inv, acc.t[3][r]);
    else acc.t[3][r] = __builtin_fmaf(lo_p[r], F2_DOWN, hi_p[r]);
  }
  __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const f16x8 ah0 = __builtin_bit_cast(f16x8, rm_frag<TR>(img, 0, t, 0)), ah1 = __builtin_bit_cast(f16x8, rm_frag<TR>(img, 0, t, 1));
    const f16x8 al0 = __builtin_bit_cast(f16x8, rm_frag<TR>(img, 1, t, 0)), al1 = __builtin_bit_cast(f16x8, rm_frag<TR>(img, 1, t, 1));
    f32x4 lo = {0.f, 0.f, 0.f, 0.f};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xh0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xl0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xh1, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xl1, lo, 0, 0, 0);
    if constexpr (SCALED) {
      f32x4 hi;
#pragma unroll
      for (int r = 0; r < 4; ++r) hi[r] = lo[r] * F2_DOWN;
      hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh0, hi, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh1, hi, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc.t[t][r] = __builtin_fmaf(hi[r], inv, acc.t[t][r]);
    } else {
      f32x4 hi = acc.t[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) hi[r] = __builtin_fmaf(lo[r], F2_DOWN, hi[r]);
      hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh0, hi, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh1, hi, 0, 0, 0);
      acc.t[t] = hi;
    }
  }
  }
}

// ---- f16x2 for the in-workgroup WEIGHT-GRADIENT consumers (round 5): sticky power-of-two scale, 32x32x16 MFMAs ----------------
// dW[o][k] += sum over the 16 rows of a ring ticket of G[row][o] T[row][k].  A contraction over ROWS cannot take the per-item scale
// of vsplit2_scaled; it takes a scale per operand STREAM instead: S = 2^k chosen so that the largest magnitude seen so far sits in
// [2^14, 2^15), lowered (never raised) when a larger tile arrives -- the accumulator is multiplied by the ratio then, a rare
// wave-uniform branch.  Errors are 2^-23 of the stream's largest row, which is what an fp32 sum over rows gives too.  Near 2^14 the
// residual x S - fp16(x S) is itself a normal fp16 number down to |x S| = 1/4 (absolute error 2^-25 below that: 2^-39 of the
// largest element), so the low part needs NO 2^11 scale and all three products -- (h,h), (l,h), (h,l) -- accumulate into ONE
// register set: 12 MFMAs and no fold per ticket and weight where the bf16x3 form took 48 MFMAs per ticket.
// v_mfma_f32_32x32x16_f16: K = 16 rows = exactly one ticket (no pairing of tickets), and the instruction holds the vector issue port
// for 8 of its 32 cycles instead of 8 of 16 (MI355X_MICROARCH.md, constants table).
// Operand layout: lane (i = lane & 31, half = lane >> 5) supplies rows 8 half + e (e = 0..7) of feature 32 b + i for block b = 0, 1
// -- of G as the A operand, of T as the B operand.  Accumulator block (bo, bk), register 4 g + r of lane (n, half):
// dW[32 bo + 8 g + 4 half + r][32 bk + n].
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct WgAcc32 {
  f32x16 c[2][2];
};
struct WgOp32 {
  u32x4 h[2], l[2];   // [block]: 8 fp16 per lane
};
__device__ __forceinline__ void wg32_zero(WgAcc32 &a) {
#pragma unroll
  for (int bo = 0; bo < 2; ++bo)
#pragma unroll
    for (int bk = 0; bk < 2; ++bk)
#pragma unroll
      for (int e = 0; e < 16; ++e) a.c[bo][bk][e] = 0.f;
}
// x[b][e] = tile[(8 half + e) * RS + 32 b + i]: conflict-free b32 reads (32 consecutive floats per half, the halves 8 rows apart)
template <int RS>
__device__ __forceinline__ void wg32_read(const float *tile, float (&x)[2][8]) {
  const int l = lane_id(), i = l & 31, hf = l >> 5;
  const float *p = tile + (8 * hf) * RS + i;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) x[b][e] = p[e * RS + 32 * b];
}
// largest magnitude of the tile, the same value in every lane (DPP row rotations, then the two permlane swaps)
__device__ __forceinline__ float wg32_absmax(const float (&x)[2][8]) {
  float m = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[b][e]));
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x128, 0xf, 0xf, false)));  // row_ror:8
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x124, 0xf, 0xf, false)));  // row_ror:4
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x122, 0xf, 0xf, false)));  // row_ror:2
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x121, 0xf, 0xf, false)));  // row_ror:1
  return qmax(m);
}
// does the tile fit under scale exponent `se`?  One lane-local maximum and a ballot -- no cross-lane reduction on the consumer's
// critical path; the full maximum (wg32_absmax) is taken only when this says no (the first tile of a stream, or a larger one).
__device__ __forceinline__ bool wg32_fits(const float (&x)[2][8], int se) {
  float m = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[b][e]));
  // |x| S < 2^15  <=>  exponent(x) + se - 127 <= 141; written on the bits: (bits(|x|) >> 23) <= 268 - se
  const bool over = (int)(f2u(m) >> 23) > 268 - se;
  return se != 0 && __builtin_amdgcn_ballot_w64(over) == 0ull;
}
// sticky scale of one operand stream: `se` = exponent field of S (0: no tile seen yet).  Returns the factor the accumulators of the
// stream must be multiplied by (1 unless the scale had to come down); the caller applies it.
struct WgScale {
  int se;
  // the common case costs a lane-local maximum and one ballot
  __device__ __forceinline__ float update_lazy(const float (&x)[2][8]) {
    if (wg32_fits(x, se)) return 1.f;
    return update(wg32_absmax(x));
  }
  __device__ __forceinline__ float scale() const { return __builtin_bit_cast(float, (unsigned)(se > 0 ? se : 127) << 23); }
  __device__ __forceinline__ float inv() const { return __builtin_bit_cast(float, (unsigned)(254 - (se > 0 ? se : 127)) << 23); }
  // m: the tile's largest magnitude (wave-uniform)
  __device__ __forceinline__ float update(float m) {
    const int em = __builtin_amdgcn_readfirstlane((int)(f2u(m) >> 23));   // biased exponent; 0 for a zero (or subnormal) tile
    if (em == 0 || em == 255) return 1.f;                                 // nothing to scale; Inf / NaN tiles poison the sum as in fp32
    int want = 268 - em;                                                  // m S in [2^14, 2^15)
    want = want > 254 ? 254 : (want < 1 ? 1 : want);
    if (se == 0) { se = want; return 1.f; }
    if (want >= se) return 1.f;                                           // the tile fits under the current scale
    const float f = __builtin_bit_cast(float, (unsigned)(127 - (se - want)) << 23);   // 2^(want - se) < 1
    se = want;
    return f;
  }
};
// the 16 values of a lane -> fp16 parts of x S: h = fp16(x S) (RNE), l = fp16(x S - h)
__device__ __forceinline__ WgOp32 wg32_split(const float (&x)[2][8], float sc) {
  WgOp32 O;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    unsigned h[4];
    mix_pack4(x[b], sc, h);   // (an MFMA operand written by inline assembly: the block ends with the wait states, see vsplit2)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a0 = x[b][2 * w], a1 = x[b][2 * w + 1];
      float r0, r1;
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(sc), "v"(h[w]));
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(sc), "v"(h[w]));
      O.h[b][w] = h[w];
      O.l[b][w] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
    }
  }
  return O;
}
__device__ __forceinline__ void wg32_scale_acc(WgAcc32 &a, float f) {
#pragma unroll
  for (int bo = 0; bo < 2; ++bo)
#pragma unroll
    for (int bk = 0; bk < 2; ++bk)
#pragma unroll
      for (int e = 0; e < 16; ++e) a.c[bo][bk][e] *= f;
}
// acc += G^T T over the 16 rows: smallest terms first
__device__ __forceinline__ void wg32_mma(WgAcc32 &acc, const WgOp32 &G, const WgOp32 &T) {
#pragma unroll
  for (int bo = 0; bo < 2; ++bo)
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) {
      f32x16 c = acc.c[bo][bk];
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, G.l[bo]), __builtin_bit_cast(f16x8, T.h[bk]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, G.h[bo]), __builtin_bit_cast(f16x8, T.l[bk]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, G.h[bo]), __builtin_bit_cast(f16x8, T.h[bk]), c, 0, 0, 0);
      acc.c[bo][bk] = c;
    }
}

// Which forward kernels run on f16x2 images (bit 0 edge_fwd, bit 1 virt_fwd, bit 2 node_pre_fwd); pack_kernel writes the
// split images those kernels read (and only they read) in the matching format.
#ifndef FE_FWD_F16
#define FE_FWD_F16 7   // adopted in round 4: cfg4 step 12.20 -> 11.90 ms on one box, parity suite green, FEWER comparisons beyond 2 x ref
#endif
// Arithmetic form of the 64x64 layers of a kernel (template parameter of the stage kernels)
enum GemmMode { GM_F32 = 0, GM_X3 = 1, GM_BF16 = 2, GM_F16 = 3 };
constexpr int GM_EDGE_FWD = (FE_FWD_F16 & 1) ? GM_F16 : GM_X3;
constexpr int GM_VIRT_FWD = (FE_FWD_F16 & 2) ? GM_F16 : GM_X3;
constexpr int GM_NODE_PRE_FWD = (FE_FWD_F16 & 4) ? GM_F16 : GM_X3;
// -DFE_BWD_F16=mask: the producers of edge_bwd_pc (bit 0) / virt_bwd_pc (bit 1) on f16x2 products (their in-workgroup weight-
// gradient consumers keep the bf16x3 form: a contraction over ROWS has no per-item scale)
#ifndef FE_BWD_F16
#define FE_BWD_F16 3   // adopted in round 4: 11.90 -> 11.45 ms (virt_bwd 3.51 -> 3.30, edge_bwd 3.39 -> 3.16); -DFE_BWD_F16=0 restores bf16x3
#endif
constexpr int GM_EDGE_BWD = (FE_BWD_F16 & 1) ? GM_F16 : GM_X3;
constexpr int GM_VIRT_BWD = (FE_BWD_F16 & 2) ? GM_F16 : GM_X3;
// the B operand of one or several products in the chosen form: made once, used by every layer that reads it
template <int MODE> struct OperandOf { typedef Vec type; };
template <> struct OperandOf<GM_X3> { typedef Split type; };
template <> struct OperandOf<GM_BF16> { typedef BfOp type; };
template <> struct OperandOf<GM_F16> { typedef Split2 type; };
// the B operand made from a GRADIENT (transposed products): the f16x2 form scales per item, the other forms do not care
template <int MODE> struct GradOperandOf { typedef typename OperandOf<MODE>::type type; };
template <> struct GradOperandOf<GM_F16> { typedef Split2s type; };
template <int MODE>
__device__ __forceinline__ typename OperandOf<MODE>::type make_operand(const Vec &v);
template <int MODE>
__device__ __forceinline__ typename GradOperandOf<MODE>::type make_grad_operand(const Vec &v) {
  if constexpr (MODE == GM_F16) return vsplit2_scaled(v);
  else return make_operand<MODE>(v);
}
template <int MODE>
__device__ __forceinline__ typename OperandOf<MODE>::type make_operand(const Vec &v) {
  if constexpr (MODE == GM_X3) return vsplit(v);
  else if constexpr (MODE == GM_BF16) return vpack_bf(v);
  else if constexpr (MODE == GM_F16) return vsplit2(v);
  else return v;
}
// ---- the 32-edge forward kernel (edge_fwd32.hip): does this build have it, and where its images live ----
// Measured, NOT adopted (profiles/r05_lever_edge_fwd32.txt): the 32-edge kernel issues 11 % fewer cycles per edge but runs two waves per
// SIMD (242-256 registers) where the 16-edge kernel runs four, and loses more to exposed MFMA / LDS latency than it saves:
// 0.837 against 0.797 ms per step on one box.  -DFE_EDGE_FWD32=1 builds it in (and makes pack_kernel write its images).
#ifndef FE_EDGE_FWD32
#define FE_EDGE_FWD32 0
#endif
#ifdef FE_ACT_GENERIC
constexpr bool EDGE_FWD32 = false;
#else
constexpr bool EDGE_FWD32 = FE_EDGE_FWD32 != 0 && (FE_FWD_F16 & 1) != 0 && LOG2E_FOLD_FIRST;
#endif
// u32 index (two fp16: contraction indices k, k + 1, k even) of W[o][k] in an f16x2 image of the 32x32x16 operand layout:
// part h at 0, part l at 2048; lane (i, hf) of output block bo, k-step s reads its 8 values as one 16-byte word group
__host__ __device__ inline int img32_word(int part, int o, int k) {
  const int bo = o >> 5, i = o & 31, b = k >> 5, g = (k >> 3) & 3, hf = (k >> 2) & 1, r = k & 3;
  const int s = 2 * b + (g >> 1), e = 4 * (g & 1) + r;
  return part * 2048 + ((bo * 4 + s) * 64 + hf * 32 + i) * 4 + (e >> 1);
}
// image i of a resident image array (fp32 images for GM_F32, split images otherwise)
template <int MODE, bool PIPE = false>
__device__ __forceinline__ void gemm_op(const void *img, int i, const typename OperandOf<MODE>::type &in, Vec &acc) {
  if constexpr (MODE == GM_X3) gemm64_x3(reinterpret_cast<const unsigned *>(img) + i * IMG3, in, acc);
  else if constexpr (MODE == GM_BF16) gemm64_b1(reinterpret_cast<const unsigned *>(img) + i * IMG3, in, acc);
  else if constexpr (MODE == GM_F16) gemm64_f2<PIPE>(reinterpret_cast<const unsigned *>(img) + i * IMG3, in, acc);
  else gemm64(reinterpret_cast<const float *>(img) + i * IMG, in, acc);
}
// a product on an fp32 image (fp32-input MFMA) inside a kernel of form MODE: in bf16 mode the activation is rounded
// (the image already holds bf16-representable weights), so the product has the bf16-mode semantics exactly
// product (TR = false) or transposed product (TR = true) on a row-major split image, forms GM_X3 / GM_BF16
// (PIPE: the hand-pipelined order of the bf16x3 form, P16: of the f16x2 form -- separate switches, the two forms differ by 32 registers)
template <int MODE, bool TR, bool PIPE = true, bool P16 = true>
__device__ __forceinline__ void gemm_rm(const char *img, const typename OperandOf<MODE>::type &in, Vec &acc) {
  static_assert(MODE == GM_X3 || MODE == GM_BF16 || MODE == GM_F16, "row-major images hold 16-bit parts");
  if constexpr (MODE == GM_X3) gemm64_x3_rm<TR, PIPE>(img, in, acc);
  else if constexpr (MODE == GM_F16) gemm64_f2_rm_<TR, false, P16>(img, in, 1.f, acc);
  else gemm64_b1_rm<TR>(img, in, acc);
}
// bytes of one row-major image as a kernel of form MODE keeps it in LDS: an f16x2 image has two parts
template <int MODE> constexpr int rm_lds_bytes() { return MODE == GM_F16 ? 2 * RM_PART : RM_BYTES; }
// the same with a gradient operand (make_grad_operand)
template <int MODE, bool TR, bool PIPE = true, bool P16 = true>
__device__ __forceinline__ void gemm_rm_g(const char *img, const typename GradOperandOf<MODE>::type &in, Vec &acc) {
  if constexpr (MODE == GM_F16) gemm64_f2_rm_<TR, true, P16>(img, in.s, in.inv, acc);
  else gemm_rm<MODE, TR, PIPE>(img, in, acc);
}
template <int MODE>
__device__ __forceinline__ void gemm64_m(const float *img, const Vec &in, Vec &acc) {
  if constexpr (MODE == GM_BF16) gemm64(img, vround(in), acc);
  else gemm64(img, in, acc);
}

// cooperative copy of n_img consecutive images global -> LDS (16-byte moves)
__device__ __forceinline__ void load_images(float *dst, const float *src, int n_img) {
  const f32x4 *s = reinterpret_cast<const f32x4 *>(src);
  f32x4 *d = reinterpret_cast<f32x4 *>(dst);
  for (int i = threadIdx.x; i < n_img * (IMG / 4); i += blockDim.x) d[i] = s[i];
}
__device__ __forceinline__ void load_images_x3(unsigned *dst, const unsigned *src, int n_img) {
  const u32x4 *s = reinterpret_cast<const u32x4 *>(src);
  u32x4 *d = reinterpret_cast<u32x4 *>(dst);
  for (int i = threadIdx.x; i < n_img * (IMG3 / 4); i += blockDim.x) d[i] = s[i];
}
__device__ __forceinline__ void load_floats(float *dst, const float *src, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src ? src[i] : 0.f;
}

// ---- transpose tile (per wave, [16][TS]) ----
__device__ __forceinline__ void tile_store(float *tile, int j, int q, const Vec &v) {
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(tile + j * TS + 16 * t + 4 * q) = v.t[t];
}

// ---- packed weight-image table (one per layer), see pack.hip ----
enum ImgId {
  I_W2 = 0, I_WX1, I_W2T, I_WX1T,                       // edge stage (fwd pair first)
  I_V2, I_WXV0, I_WXX0, I_V2T, I_WXV0T, I_WXX0T,         // virtual stage
  I_W1A, I_W1B, I_V1A, I_WVEL0, I_WG0,                   // node_pre forward
  I_W1AT, I_W1BT, I_V1AT, I_WVEL0T, I_WG0T,              // node_pre backward
  I_W5A, I_W5B, I_W6, I_W5AT, I_W5BT, I_W6T,             // graph_post
  I_V1B, I_V1BT,                                         // graph_pre
  I_W3A, I_W3B, I_W4, I_W3AT, I_W3BT, I_W4T,             // node MLP
  I_FIXED,                                               // then W3c[0..C) and W3cT[0..C)
};
__host__ __device__ inline int img_w3c(int c) { return I_FIXED + c; }
// split image `id` is written as an f16x2 image (parts h | l, FE_FWD_F16) instead of the bf16 h | m | l parts
__host__ __device__ inline bool img_is_f16(int id, int C) {
  if (id == I_W2 || id == I_WX1) return (FE_FWD_F16 & 1) != 0;
  if (id == I_V2 || id == I_WXV0 || id == I_WXX0 || (id >= I_FIXED && id < I_FIXED + C)) return (FE_FWD_F16 & 2) != 0;
  if (id == I_W3A || id == I_W3B || id == I_W4) return (FE_FWD_F16 & 2) != 0;   // node_model's three products at the tail of virt_fwd
  if (id >= I_W1A && id <= I_WG0) return (FE_FWD_F16 & 4) != 0;
  if (id >= I_FIXED + C && id < I_FIXED + 2 * C) return (FE_BWD_F16 & 2) != 0;   // W3c[c]^T: read by virt_bwd_cs_kernel only
  return false;
}
__host__ __device__ inline int img_w3ct(int C, int c) { return I_FIXED + C + c; }
// wpack = [fp32 images n x 4096 floats][split images n x IMG3 words (h | m | l)][row-major split images (5 + C) x RM_WORDS]
// row-major images: slot 0 V2, 1 WXV0, 2 WXX0 (virt_bwd), 3 W2, 4 WX1 (edge_bwd), 5 WVEL0, 6 WG0 (node_pre_bwd), 7..11 f16x2 forms of 0..4,
// RM_FIXED + c: W3c[c] (virtual backward)
// 7..11: the f16x2 forms (parts h | l, part 2 unused) of slots 0..4, for the backward producers built with FE_BWD_F16
constexpr int RM_F16 = 7;
constexpr int RM_FIXED = 12;
__host__ __device__ inline size_t wpack_images(int C) { return (size_t)(I_FIXED + 2 * C); }
__host__ __device__ inline size_t wpack_rm_images(int C) { return (size_t)(RM_FIXED + (C > 0 ? C : 0)); }
__host__ __device__ inline size_t wpack_floats(int C) { return wpack_images(C) * (IMG + IMG3) + wpack_rm_images(C) * RM_WORDS; }
__host__ __device__ inline int rm_slot(int id) {   // -1: the image has no row-major copy
  return id == I_V2 ? 0 : id == I_WXV0 ? 1 : id == I_WXX0 ? 2 : id == I_W2 ? 3 : id == I_WX1 ? 4 : id == I_WVEL0 ? 5 : id == I_WG0 ? 6 : -1;
}
__host__ __device__ inline const char *wpack_rm(const float *wpack, int C, int slot) {
  return reinterpret_cast<const char *>(wpack + wpack_images(C) * (IMG + IMG3)) + (size_t)slot * RM_BYTES;
}
__host__ __device__ inline const unsigned *wpack_x3(const float *wpack, int C, int id) {
  return reinterpret_cast<const unsigned *>(wpack + wpack_images(C) * IMG) + (size_t)id * IMG3;
}

}  // namespace fe
