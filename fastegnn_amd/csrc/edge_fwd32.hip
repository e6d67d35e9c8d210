// S3 edge stage, forward, on 32-edge tiles (round 5).  Same math as edge_fwd_kernel (layer_fwd.hip; reference:
// models/FastEGNN.py:102-108,125-133,180-189 -- coord2radial + edge_model + the real part of coord_model_vel + the segment means),
// other tile shape: a wave works on 32 edges and multiplies with v_mfma_f32_32x32x16_f16.
//
// Why: the 16-edge kernel is bound by vector ISSUE, and an MFMA holds the issue port for 8 cycles whatever its size
// (MI355X_MICROARCH.md, constants table).  profiles/r05_edge_fwd_instruction_budget.txt: 52 MFMAs per 16 edges = 416 of 2490 issue
// cycles.  A 32x32x16 MFMA does four times the work of a 16x16x32 in twice its matrix-pipe time: 28 MFMAs per 32 edges (112 issue
// cycles per 16), the A fragments of a layer are read from LDS once per 32 edges, and the per-item scalar work (geometry, head
// dot, walk head) is issued once per 32 edges instead of once per 16.
//
// Layout ("D32"): lane = 32 hf + n, n = edge of the tile, hf = 0 / 1.  The 64 hidden features of edge n are held by its two lanes
// as 2 x f32x16:  c[b][4 g + r]  ==  feature 32 b + 8 g + 4 hf + r  -- the C / D fragment of a 32x32 MFMA with the weights as the A
// operand (rows 32 b ..) and the edges as the B columns.  The B operand of k-step s (16 of the 64 contraction indices) takes from
// lane (n, hf) the eight values (b = s >> 1, g = 2 (s & 1) + (e >> 2), r = e & 3) -- eight of the values the lane holds -- so the
// output of one layer is the operand of the next, as in the 16-wide layout; the weight images are permuted to match
// (img32_index, written by pack_kernel into the otherwise unused fp32 image slots of W2 / WX1).
#include "stages.h"

namespace fe {

#ifndef FE_EDGE32_WAVES
#define FE_EDGE32_WAVES 8
#endif
constexpr int E32_WAVES = FE_EDGE32_WAVES;
constexpr int E32_TS = 68;                       // row stride of the [32][64] transpose tile
constexpr int E32_IMG_WORDS = 4096;              // one f16x2 image: parts h | l, 2048 words each

struct V32 {
  f32x16 c[2];
};
struct Split32 {
  u32x4 h[4], l[4];   // [k-step]: 8 fp16 per lane
};
template <typename F>
__device__ __forceinline__ V32 v32_map(const V32 &a, F f) {
  V32 o;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) o.c[b][e] = f(a.c[b][e]);
  return o;
}
// natural-order 64-vector in LDS -> this lane's 32 elements
__device__ __forceinline__ V32 v32_load_vec(const float *w, int hf) {
  V32 v;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 x = *reinterpret_cast<const f32x4 *>(w + 32 * b + 8 * g + 4 * hf);
#pragma unroll
      for (int r = 0; r < 4; ++r) v.c[b][4 * g + r] = x[r];
    }
  return v;
}
// row of a [*, ld] table (wave-uniform base, 32-bit element offset of the row start + 4 hf)
__device__ __forceinline__ V32 v32_load_u(const float *base, unsigned off) {
  V32 v;
  const char *p = reinterpret_cast<const char *>(base);
  const unsigned bo = off * 4u;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 x = *reinterpret_cast<const f32x4 *>(p + (bo + 128u * b + 32u * g));
#pragma unroll
      for (int r = 0; r < 4; ++r) v.c[b][4 * g + r] = x[r];
    }
  return v;
}
__device__ __forceinline__ Split32 v32_split(const V32 &v) {
  Split32 S;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e = 2 * w, b = s >> 1, g = 2 * (s & 1) + (e >> 2), r = e & 3;
      unsigned ph, pl;
      part2_pack(v.c[b][4 * g + r], v.c[b][4 * g + r + 1], ph, pl);
      S.h[s][w] = ph;
      S.l[s][w] = pl;
    }
  return S;
}
// acc += W x on an f16x2 image in the 32x32x16 operand layout (LDS).  K-STEP outer: the products of k-steps 2 b, 2 b + 1 need only block b
// of the input, so the vector work that produces block 1 (its SiLU, its split) is independent of the MFMAs of block 0 and the
// scheduler can run it in their shadow -- a 32x32x16 chain is 32 cycles per instruction, and with two or three waves per SIMD there
// is nobody else to fill them.  Both output blocks accumulate side by side: lo[bo] (cross products, carries the 2^11) and hi[bo];
// the fold lo -> hi comes once, at the end.
struct Acc32 {
  f32x16 lo[2], hi[2];
};
__device__ __forceinline__ void gemm32_begin(Acc32 &A, const V32 &bias) {
#pragma unroll
  for (int bo = 0; bo < 2; ++bo) {
    A.hi[bo] = bias.c[bo];
#pragma unroll
    for (int e = 0; e < 16; ++e) A.lo[bo][e] = 0.f;
  }
}
// the two k-steps that read block b of the input (its 16 values of this lane)
__device__ __forceinline__ void gemm32_block(const unsigned *img, int b, const f32x16 &x, Acc32 &A) {
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(img) + lane_id();
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int s = 2 * b + sl;
    u32x4 xh, xl;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e = 2 * w, g = 2 * sl + (e >> 2), r = e & 3;
      unsigned ph, pl;
      part2_pack(x[4 * g + r], x[4 * g + r + 1], ph, pl);
      xh[w] = ph;
      xl[w] = pl;
    }
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) {
      const f16x8 ah = __builtin_bit_cast(f16x8, ip[(bo * 4 + s) * 64]), al = __builtin_bit_cast(f16x8, ip[512 + (bo * 4 + s) * 64]);
      A.lo[bo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, xh), A.lo[bo], 0, 0, 0);
      A.lo[bo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, xl), A.lo[bo], 0, 0, 0);
      A.hi[bo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, xh), A.hi[bo], 0, 0, 0);
    }
  }
}
__device__ __forceinline__ f32x16 gemm32_end(const Acc32 &A, int bo) {
  f32x16 o = A.hi[bo];
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = __builtin_fmaf(A.lo[bo][e], F2_DOWN, o[e]);
  return o;
}
__device__ __forceinline__ f32x16 silu2_16(const f32x16 &z) {
  f32x16 o;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = silu2_f(z[e]);
  return o;
}
// <v, w> over the hidden dimension: both lanes of the edge get the full dot product
__device__ __forceinline__ float v32_dot(const V32 &v, const V32 &w) {
  float p = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) p = __builtin_fmaf(v.c[b][e], w.c[b][e], p);
  float a = p, c = p;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));   // (as qsum: a + c = p[l] + p[l ^ 32] in every lane)
  return a + c;
}

struct E32Idx {
  int row, col;
  float eav[8];
};
__device__ __forceinline__ void e32_load_idx(const EdgeArgs &a, int e, E32Idx &I) {
  const unsigned eo = (unsigned)e * 4u;
  I.row = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(a.erow) + eo);
  I.col = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(a.col) + eo);
#pragma unroll
  for (int k = 0; k < 8; ++k) I.eav[k] = 0.f;
  if (a.ea_dim == 2) {
    const float2 v = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(a.ea) + 2u * eo);
    I.eav[0] = v.x;
    I.eav[1] = v.y;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < a.ea_dim) I.eav[k] = a.ea[(size_t)e * a.ea_dim + k];
  }
}

// LDS: W2 | WX1 images (32 KB) | EV_COUNT vectors | per wave [32][E32_TS] + [32][4]
__global__ __launch_bounds__(64 * E32_WAVES) void edge_fwd32_kernel(EdgeArgs a, int C) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned *img = reinterpret_cast<unsigned *>(lds);
  float *vec = lds + 2 * E32_IMG_WORDS;
  float *tiles = vec + EV_COUNT * H;
  {
    // the 32x32x16 images live in the fp32 image slots of W2 / WX1 (consecutive ids): pack.hip
    const f32x4 *s = reinterpret_cast<const f32x4 *>(a.wpack + (size_t)I_W2 * IMG);
    f32x4 *d = reinterpret_cast<f32x4 *>(lds);
    for (int i = threadIdx.x; i < 2 * E32_IMG_WORDS / 4; i += blockDim.x) d[i] = s[i];
  }
  // (written for the fully folded SiLU chain: P / Q, every pre-activation and every activation in units of ln 2 -- common.h, FE_LOG2E_FOLD)
  edge_load_vecs(vec, a, true);
  __syncthreads();
  const int l = lane_id(), n = l & 31, hf = l >> 5, wv = wave_id();
  float *mt = tiles + wv * (32 * E32_TS + 128);
  float *xt = mt + 32 * E32_TS;
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool mean = !(a.flags & FASTEGNN_F_COORDS_SUM);
  const int c0 = (int)((long)wave * a.n_chunks / nwaves), c1 = (int)((long)(wave + 1) * a.n_chunks / nwaves);
  const int r0 = a.chunk_row[c0], r1 = a.chunk_row[c1];
  if (r0 >= r1) return;
  const int e0 = a.rowptr[r0], e1 = a.rowptr[r1];
  int cur = -1, cnt = 0;
  float acc = 0.f, accx = 0.f;
  auto flush = [&]() {
    const float inv = rcp_f((float)cnt);
    a.aggm[(size_t)cur * H + l] = acc * (inv * LN2_F);
    if (l < 3) a.aggx[(size_t)cur * 3 + l] = mean ? accx * inv : accx;
  };
  auto zero_rows = [&](int ra, int rb) {
    for (int r = ra; r < rb; ++r) {
      a.aggm[(size_t)r * H + l] = 0.f;
      if (l < 3) a.aggx[(size_t)r * 3 + l] = 0.f;
    }
  };
  const float attb = (a.flags & FASTEGNN_F_ATTENTION) ? a.attb[0] : 0.f;
  const float bx2 = a.bx2 ? a.bx2[0] : 0.f;
  // Software pipeline over tiles: the indices of tile k + 2 and the gathered rows of tile k + 1 are in flight while tile k is
  // computed (72 registers: the kernel runs two waves per SIMD at <= 256 registers and hides its own gather latency -- measured:
  // without the row prefetch the kernel is latency-bound, 0.886 ms per step at two waves per SIMD, 0.809 at three).
  // -DFE_EDGE32_PREFETCH=0: rows requested at the head of their tile.
#ifndef FE_EDGE32_PREFETCH
#define FE_EDGE32_PREFETCH 1
#endif
  struct Rows {
    V32 p, q;
    f32x4 xr, xc;
  };
  auto gather = [&](const E32Idx &I, Rows &G) {
    const unsigned qoff = (unsigned)I.col * QXLD, roff = (unsigned)I.row * QXLD;
    G.xc = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(a.QXs) + (qoff + H) * 4u);
    G.xr = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(a.QX) + (roff + H) * 4u);
    G.p = v32_load_u(a.P, (unsigned)I.row * H + 4 * hf);
    G.q = v32_load_u(a.QXs, qoff + 4 * hf);
  };
  E32Idx cur_i, nxt_i, nn_i;
  Rows G, Gn;
  e32_load_idx(a, min(e0 + n, e1 - 1), cur_i);
  nxt_i = cur_i;
  if (e0 + 32 < e1) e32_load_idx(a, min(e0 + 32 + n, e1 - 1), nxt_i);
  gather(cur_i, G);
  for (int base = e0; base < e1; base += 32) {
    const int nvalid = min(32, e1 - base);
    nn_i = nxt_i;
#if FE_EDGE32_PREFETCH
    if (base + 64 < e1) e32_load_idx(a, min(base + 64 + n, e1 - 1), nn_i);
    if (base + 32 < e1) gather(nxt_i, Gn);     // next tile's rows: consumed in the next trip
#else
    if (base + 64 < e1) e32_load_idx(a, min(base + 64 + n, e1 - 1), nn_i);
    if (base > e0) gather(cur_i, G);
#endif
    // ---- geometry + first-layer sum
    const f32x4 xc = G.xc, xr = G.xr;
    V32 pre = G.p;
#pragma unroll
    for (int b = 0; b < 2; ++b) pre.c[b] += G.q.c[b];
    float d[3] = {xr[0] - xc[0], xr[1] - xc[1], xr[2] - xc[2]};
    const float r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (a.flags & FASTEGNN_F_NORMALIZE) {
      const float inv = rcp_f(sqrt_f(r2) + a.eps);
      d[0] *= inv; d[1] *= inv; d[2] *= inv;
    }
    // ---- first layer: the rank-(1 + ea_dim) update on the matrix pipe, K = 2 scalar features per v_mfma_f32_32x32x2_f32: lane
    // (i, hf) supplies row 2 st + hf of the feature weights (vec rows: EV_WR = 0, EV_WE + k = 1 + k) for output 32 b + i, lane (n, hf)
    // feature 2 st + hf of its edge
    {
      const float rf = (a.flags & FASTEGNN_F_EGNN_NORM) ? (r2 >= 1e-12f ? 1.0f : r2 * 1e12f) : r2;
      const int nst = (a.ea_dim + 2) >> 1;   // features: radial, edge_attr[0 .. ea_dim)
      float f[4];
      f[0] = hf ? cur_i.eav[0] : rf;
      f[1] = hf ? cur_i.eav[2] : cur_i.eav[1];
      f[2] = hf ? cur_i.eav[4] : cur_i.eav[3];
      f[3] = hf ? cur_i.eav[6] : cur_i.eav[5];
#pragma unroll
      for (int st = 0; st < 4; ++st)
        if (st < nst) {   // wave-uniform
          const float fs = f[st] * LOG2E_F;
          const float *w = vec + (2 * st + hf) * H + n;
#pragma unroll
          for (int b = 0; b < 2; ++b) pre.c[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[32 * b], fs, pre.c[b], 0, 0, 0);
        }
    }
    // ---- the two 64x64 layers and the coordinate head (everything downstream of `pre` in units of ln 2: common.h, FE_LOG2E_FOLD)
    Acc32 A;
    gemm32_begin(A, v32_load_vec(vec + EV_B2 * H, hf));
#pragma unroll
    for (int b = 0; b < 2; ++b) gemm32_block(img, b, silu2_16(pre.c[b]), A);     // t = silu(pre), block by block
    V32 m;
    if (a.flags & FASTEGNN_F_ATTENTION) {
#pragma unroll
      for (int b = 0; b < 2; ++b) m.c[b] = silu2_16(gemm32_end(A, b));
      const float att = sigmoid_f(v32_dot(m, v32_load_vec(vec + EV_ATT * H, hf)) + attb);
      m = v32_map(m, [att](float z) { return z * att; });
      gemm32_begin(A, v32_load_vec(vec + EV_BX1 * H, hf));
#pragma unroll
      for (int b = 0; b < 2; ++b) gemm32_block(img + E32_IMG_WORDS, b, m.c[b], A);
    } else {
      // block 0 of the message feeds k-steps 0, 1 of the coordinate head's first layer while block 1 is still being finished
      Acc32 B;
      gemm32_begin(B, v32_load_vec(vec + EV_BX1 * H, hf));
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        m.c[b] = silu2_16(gemm32_end(A, b));
        gemm32_block(img + E32_IMG_WORDS, b, m.c[b], B);
      }
      A = B;
    }
    V32 u;
#pragma unroll
    for (int b = 0; b < 2; ++b) u.c[b] = silu2_16(gemm32_end(A, b));
    const float sraw = v32_dot(u, v32_load_vec(vec + EV_WX2 * H, hf)) + bx2;
    const float s = (a.flags & FASTEGNN_F_TANH) ? tanh_f(sraw) : sraw;
    // ---- transpose tile: m rows [32][E32_TS], coordinate parts [32][4]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = m.c[b][4 * g + r];
        *reinterpret_cast<f32x4 *>(mt + n * E32_TS + 32 * b + 8 * g + 4 * hf) = x;
      }
    if (hf == 0) *reinterpret_cast<f32x4 *>(xt + n * 4) = f32x4{d[0] * s, d[1] * s, d[2] * s, 0.f};
    __builtin_amdgcn_wave_barrier();
    // ---- row walk: hidden-on-lane column of the tile, sixteen edges at a time
    const int rowv = cur_i.row;
    const int prevrow = __builtin_amdgcn_update_dpp(rowv, rowv, 0x111, 0xf, 0xf, false);   // row_shr:1 (lane 0 of a 16-lane row keeps its own)
    // lane n compares with lane n - 1; lanes 0 and 16 (first of their DPP row) with the walk's running row / with lane 15
    const int row15 = __builtin_amdgcn_readlane(rowv, 15);
    const bool chgl = n == 0 ? rowv != cur : (n == 16 ? rowv != row15 : rowv != prevrow);
    const unsigned starts = (unsigned)__builtin_amdgcn_ballot_w64(chgl);   // bits 0..31: the edges of the tile (hf = 0 lanes)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float mv[16], xv[16];
#pragma unroll
      for (int ee = 0; ee < 16; ++ee) {
        mv[ee] = mt[(16 * half + ee) * E32_TS + l];
        xv[ee] = xt[(16 * half + ee) * 4 + (l & 3)];
      }
#pragma unroll
      for (int ee = 0; ee < 16; ++ee) {
        const int k = 16 * half + ee;
        if (k < nvalid) {
          if ((starts >> k) & 1u) {
            const int rw = __builtin_amdgcn_readlane(rowv, k);
            if (cur >= 0) flush();
            zero_rows(cur >= 0 ? cur + 1 : r0, rw);
            cur = rw;
            acc = 0.f;
            accx = 0.f;
            cnt = 0;
          }
          acc += mv[ee];
          accx += xv[ee];
          ++cnt;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    cur_i = nxt_i;
    nxt_i = nn_i;
#if FE_EDGE32_PREFETCH
    G = Gn;
#endif
  }
  if (cur >= 0) flush();
  zero_rows(cur >= 0 ? cur + 1 : r0, r1);
}

// is the 32-edge kernel the one that runs this layer's edge stage?  (the default build in fp32-grade SiLU mode; FASTEGNN_EDGE_FWD32=0
// in the environment keeps the 16-edge kernel -- an A/B switch)
bool edge_forward32_applies(const fastegnn_layer_t *L) {
  if (!EDGE_FWD32) return false;
  static const bool off = getenv("FASTEGNN_EDGE_FWD32") && atoi(getenv("FASTEGNN_EDGE_FWD32")) == 0;
  return !off && !has(L, FASTEGNN_F_BF16);
}

int edge_forward32(const fastegnn_layer_t *L, hipStream_t st) {
  const fastegnn_graph_t &g = L->graph;
  EdgeArgs a = make_edge_args(L);
  // one workgroup per CU once there are enough 32-edge row chunks; small graphs spread their chunks over as many waves as there are chunks
  int grid = cdiv(g.n_chunks, E32_WAVES);
  if (grid > 256) grid = 256;
  const size_t lds = (2 * E32_IMG_WORDS + EV_COUNT * H + E32_WAVES * (32 * E32_TS + 128)) * sizeof(float);
  {
    ProfScope ps(K_EDGE_FWD, st);
    hipLaunchKernelGGL(edge_fwd32_kernel, dim3(grid), dim3(64 * E32_WAVES), lds, st, a, L->C);
  }
  return check_launch("edge_fwd32_kernel");
}

}  // namespace fe
