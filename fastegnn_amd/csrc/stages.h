// Device code shared by the forward and backward stage kernels (the backward recomputes the
// forward of a tile instead of storing [E,64] / [N,C,64] activations).
#pragma once
#include "kernels.h"

namespace fe {

// The 64x64 layers of the edge / virtual kernels read split (bf16) weight images and run on the matrix pipe: as
// bf16x3 products (fp32-grade, GM_X3) or, in the bf16 operand mode, as one bf16 product (GM_BF16); common.h.
#ifndef FE_EDGE_FWD_WAVES
#define FE_EDGE_FWD_WAVES 16
#endif
constexpr int EDGE_FWD_WAVES = FE_EDGE_FWD_WAVES;   // 48 KB of split images are shared by more waves
#ifndef FE_NODE_PRE_WAVES
#define FE_NODE_PRE_WAVES 8
#endif
constexpr int NODE_PRE_WAVES = FE_NODE_PRE_WAVES;   // 8: two waves per SIMD (<= 256 registers each)
#ifndef FE_VIRT_WAVES
#define FE_VIRT_WAVES 8
#endif
constexpr int VIRT_WAVES = FE_VIRT_WAVES;
#ifndef FE_VIRT_BWD_WAVES
#define FE_VIRT_BWD_WAVES 4
#endif
constexpr int VIRT_BWD_WAVES = FE_VIRT_BWD_WAVES;   // 1 wave/SIMD: the adjoint of the virtual block needs > 256 registers

struct EdgeArgs {
  const float *P, *QX, *QXs, *ea, *wpack;
  const float *E0W, *b2, *bx1, *wx2, *attw, *attb, *bx2;   // bx2: coordinate-head bias (EGNN baseline) or null
  const int32_t *rowptr, *erow, *col, *chunk_row;
  float *aggm, *aggx;
  int n_chunks, ea_dim, flags;
  float eps;
  float act_param = 0.f;   // parameter of the activation kind in the flags (generic-activation build only)
};

constexpr int EV_WR = 0, EV_WE = 1, EV_B2 = 9, EV_BX1 = 10, EV_WX2 = 11, EV_ATT = 12, EV_COUNT = 13;

// fold: the forward kernel's log2(e) fold (common.h, FE_LOG2E_FOLD): biases of the two in-kernel products in units of ln 2,
// the vectors that read their activations divided by log2(e)
__device__ __forceinline__ void edge_load_vecs(float *vec, const EdgeArgs &a, bool fold = false) {
  const int ld = 2 * H + 1 + a.ea_dim;
  const float up = fold ? LOG2E_F : 1.f, dn = fold ? LN2_F : 1.f;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    vec[EV_WR * H + i] = a.E0W[(size_t)i * ld + ((a.flags & FASTEGNN_F_EGNN) ? 0 : 2 * H)];
    for (int k = 0; k < 8; ++k) vec[(EV_WE + k) * H + i] = k < a.ea_dim ? a.E0W[(size_t)i * ld + 2 * H + 1 + k] : 0.f;
    vec[EV_B2 * H + i] = a.b2[i] * up;
    vec[EV_BX1 * H + i] = a.bx1[i] * up;
    vec[EV_WX2 * H + i] = a.wx2[i] * dn;
    vec[EV_ATT * H + i] = a.attw ? a.attw[i] * dn : 0.f;
  }
}

struct EdgeFwdState {
  Vec t, mp, m0, m, up, u;
  float d[3], dn[3], r, rf, nrm, att, s, eav[8];   // rf: the radial FEATURE of the message MLP (r, or its normalised form)
  int row, col;
};

// per-edge indices and scalar attributes; loaded one tile ahead of their use so that only one
// level of the (index -> gathered row) dependent-load chain is exposed per tile
struct EdgeIdx {
  int row, col;
  float eav[8];
};
__device__ __forceinline__ void edge_load_idx(const EdgeArgs &a, int e, EdgeIdx &I) {
  const unsigned eo = (unsigned)e * 4u;
  I.row = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(a.erow) + eo);
  I.col = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(a.col) + eo);
  // ea_dim is wave-uniform: scalar branches, and no select on a loaded value (a select would make the
  // load wait at issue).  Slots k >= ea_dim are zero (the backward's weight-column sums run unguarded over them).
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

// gathered operands of one 16-edge tile (requested at the head of the tile from indices that were requested one tile
// earlier: one of the two dependent round trips is exposed, the other waves of the SIMD cover it -- requesting the rows
// a tile ahead as well costs 40 registers, i.e. a wave per SIMD, and measured slower in both edge kernels)
struct EdgeRows {
  Vec p, qv;        // P[row], Q[col] in D layout
  f32x4 xr, xc;     // coordinates of the two end points
};
// (32-bit element offsets from wave-uniform bases: the launchers require the tables to stay below 2^30 floats)
__device__ __forceinline__ void edge_gather(const EdgeArgs &a, const EdgeIdx &I, int q, EdgeRows &G) {
  const unsigned qoff = (unsigned)I.col * QXLD, roff = (unsigned)I.row * QXLD;
  G.xc = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(a.QXs) + (qoff + H) * 4u);
  G.xr = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(a.QX) + (roff + H) * 4u);
  G.p = vload_u(a.P, (unsigned)I.row * H + 4 * q);
  G.qv = vload_u(a.QXs, qoff + 4 * q);
}

// forward math of one 16-edge tile (shared with the backward kernel for recomputation).
// KEEP_D: pre, S.mp and S.up return silu'(.) of the pre-activations instead of the pre-activations
// (the adjoint needs only the derivatives; one sigmoid serves both).
// MODE (GemmMode): GM_F32 = fp32 images / fp32-input MFMA; GM_X3 = split images, bf16x3 products on the matrix pipe;
// GM_BF16 = split images of bf16-rounded weights, one bf16 product of the RNE-rounded activation (FASTEGNN_F_BF16).
template <int MODE>
__device__ __forceinline__ void gemm_i(const void *img, int i, const Vec &in, Vec &acc) {
  gemm_op<MODE>(img, i, make_operand<MODE>(in), acc);
}
// The products of virt_tile_forward are issued at wave priority 3 (0 elsewhere): of the two waves of a SIMD the one that has
// matrix work gets the issue port first and the other fills the gaps with its vector work -- virt_fwd 1.64 -> 1.58 ms per
// step on one box (priority 1: 1.59).  -DVF_PRIO=0 switches it off; -DEF_PRIO=n / -DVB_PRIO=n: the same experiment in the
// edge kernels / the producers of virt_bwd_pc (measured, not adopted: edge_bwd 3.44 -> 3.61 ms at priority 2, edge_fwd and
// virt_bwd_pc unchanged at priorities 1 and 2).
#ifndef VF_PRIO
#define VF_PRIO 3
#endif
#if VF_PRIO > 0
#define VF_PRIO_ON() __builtin_amdgcn_s_setprio(VF_PRIO)
#define VF_PRIO_OFF() __builtin_amdgcn_s_setprio(0)
#else
#define VF_PRIO_ON()
#define VF_PRIO_OFF()
#endif
#ifndef FE_EB_P16   // the software-pipelined f16x2 product in the producers of edge_bwd_pc (common.h); 0 in the three-waves-per-SIMD experiment
#define FE_EB_P16 1
#endif
#ifdef EF_PRIO
#define EF_PRIO_ON() __builtin_amdgcn_s_setprio(EF_PRIO)
#define EF_PRIO_OFF() __builtin_amdgcn_s_setprio(0)
#else
#define EF_PRIO_ON()
#define EF_PRIO_OFF()
#endif
// the edge stage's four products I = 0 W2, 1 WX1, 2 W2^T, 3 WX1^T: from four split images (RM = false: img + I) or from the
// two row-major images W2 | WX1 read plain or transposed (RM = true, edge_bwd: half the LDS, which its rings take)
template <int MODE, int I, bool RM>
__device__ __forceinline__ void gemm_e(const void *img, const Vec &in, Vec &acc) {
  if constexpr (RM && I >= 2) {   // the transposed products take gradients: the f16x2 form scales them per item
    const auto op = make_grad_operand<MODE>(in);
    EF_PRIO_ON();
    gemm_rm_g<MODE, true, false, FE_EB_P16 != 0>(static_cast<const char *>(img) + (I & 1) * rm_lds_bytes<MODE>(), op, acc);
    EF_PRIO_OFF();
  } else {
    const auto op = make_operand<MODE>(in);
    EF_PRIO_ON();
    if constexpr (RM) gemm_rm<MODE, false, false, FE_EB_P16 != 0>(static_cast<const char *>(img) + (I & 1) * rm_lds_bytes<MODE>(), op, acc);
    else gemm_op<MODE, true>(img, I, op, acc);   // (edge_fwd: the software-pipelined f16x2 product, common.h)
    EF_PRIO_OFF();
  }
}

// part 1: consumes the gathered operands (geometry + first-layer pre-activation)
// FOLD1: P and Q arrive in units of ln 2 (common.h, FE_LOG2E_FOLD bit 1) -- the scalar features of the rank-3 update are scaled to match
template <bool FOLD1>
__device__ __forceinline__ void edge_tile_pre(const EdgeArgs &a, const float *vec, const EdgeIdx &I, const EdgeRows &G,
                                              int q, EdgeFwdState &S, Vec &pre FE_TP) {
  S.row = I.row;
  S.col = I.col;
  S.d[0] = G.xr[0] - G.xc[0];
  S.d[1] = G.xr[1] - G.xc[1];
  S.d[2] = G.xr[2] - G.xc[2];
  S.r = S.d[0] * S.d[0] + S.d[1] * S.d[1] + S.d[2] * S.d[2];
  S.nrm = sqrt_f(S.r);
  if (a.flags & FASTEGNN_F_NORMALIZE) {
    const float inv = rcp_f(S.nrm + a.eps);
    S.dn[0] = S.d[0] * inv; S.dn[1] = S.d[1] * inv; S.dn[2] = S.d[2] * inv;
  } else {
    S.dn[0] = S.d[0]; S.dn[1] = S.d[1]; S.dn[2] = S.d[2];
  }
  FE_T(0)   // indices + coordinates arrived
  pre = G.p;
  vadd(pre, G.qv);
  // EGNN(norm=True): F.normalize of the 1x1 Gram (basic.py:271-272, eps 1e-12)
  S.rf = (a.flags & FASTEGNN_F_EGNN_NORM) ? (S.r >= 1e-12f ? 1.0f : S.r * 1e12f) : S.r;
#pragma unroll
  for (int k = 0; k < 8; ++k) S.eav[k] = I.eav[k];
#ifdef FE_EDGE_PRE_VALU   // rounds 1-3: one vector fma per element and scalar feature
  static_assert(!FOLD1, "FE_EDGE_PRE_VALU is the unfolded form");
  vaxpy(pre, S.rf, vload_vec(vec + EV_WR * H, q));
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < a.ea_dim) vaxpy(pre, S.eav[k], vload_vec(vec + (EV_WE + k) * H, q));
#else
  // The rank-(1 + ea_dim) update pre[o][e] += sum_k Wf[o][k] * feat[k][e], feat = (radial | edge_attr), on the matrix pipe:
  // K = 4 features per v_mfma_f32_16x16x4_f32 (fp32 products, as the vector form).  Lane (q, j) supplies Wf[16t + j][4s + q]
  // as the A operand of block t (rows 4s + q of `vec`: EV_WR = 0, EV_WE + k = 1 + k; rows past the last feature are zero)
  // and feature 4s + q of its own edge as B; `pre` is the accumulator.  48 vector fmas and 12 LDS reads of a 16-edge tile
  // at edge_attr_nf = 2 become 3 selects, 4 LDS words and 4 MFMAs on a pipe that is a quarter busy.
  static_assert(EV_WR == 0 && EV_WE == 1, "feature rows of vec");
  {
    const int j = lane_id() & 15;
    // (selects of VALUES: a conditional expression over the struct's members is an lvalue, and the select of addresses it
    // becomes keeps the whole EdgeFwdState in scratch memory)
    const float r_ = S.rf, e0 = I.eav[0], e1 = I.eav[1], e2 = I.eav[2];
    float f0 = e2;
    f0 = q == 2 ? e1 : f0;
    f0 = q == 1 ? e0 : f0;
    f0 = q == 0 ? r_ : f0;
    if constexpr (FOLD1) f0 *= LOG2E_F;
    const float *w0 = vec + q * H + j;
#pragma unroll
    for (int t = 0; t < 4; ++t) pre.t[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[16 * t], f0, pre.t[t], 0, 0, 0);
    if (a.ea_dim > 3) {   // wave-uniform
      const float e3 = I.eav[3], e4 = I.eav[4], e5 = I.eav[5], e6 = I.eav[6];
      float f1 = e6;
      f1 = q == 2 ? e5 : f1;
      f1 = q == 1 ? e4 : f1;
      f1 = q == 0 ? e3 : f1;
      if constexpr (FOLD1) f1 *= LOG2E_F;
      const float *w1 = w0 + 4 * H;
#pragma unroll
      for (int t = 0; t < 4; ++t) pre.t[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[16 * t], f1, pre.t[t], 0, 0, 0);
    }
  }
#endif
  FE_T(1)   // gathered rows arrived, pre-activation formed
}
// part 2: the two 64x64 layers and the coordinate head
// does this instantiation run with the log2(e) fold?  (the forward kernel in an fp32-grade form of the SiLU build only)
template <bool KEEP_D, int MODE, bool RM>
constexpr bool edge_fold() {
  return !KEEP_D && !RM && (MODE == GM_X3 || MODE == GM_F16) && LOG2E_FOLD_EDGE;
}
// are P and Q stored in units of ln 2?  (every fp32-grade form of the SiLU build; pack.hip and node_pre_fwd decide by the same rule)
template <int MODE>
constexpr bool edge_fold_first() { return (MODE == GM_X3 || MODE == GM_F16) && LOG2E_FOLD_FIRST; }
template <bool KEEP_D, int MODE = GM_F32, bool RM = false>
__device__ __forceinline__ void edge_tile_mlp(const EdgeArgs &a, const void *img, const float *vec, int q, EdgeFwdState &S,
                                              Vec &pre FE_TP) {
  constexpr bool FOLD = edge_fold<KEEP_D, MODE, RM>();   // S.mp, S.m0, S.m, S.up, S.u are then log2(e) x their values
  constexpr bool FOLD1 = edge_fold_first<MODE>();        // pre arrives as log2(e) x the pre-activation; forward: S.t is log2(e) x t too
  static_assert(!FOLD1 || KEEP_D || FOLD, "a folded first layer needs the folded forward chain");
  if constexpr (FOLD1) S.t = KEEP_D ? vsilu_keep_d2(pre) : vsilu2(pre);
  else S.t = KEEP_D ? vsilu_keep_d(pre FE_ACT(a)) : vsilu(pre FE_ACT(a));
  FE_T(2)   // silu 1
  S.mp = vload_vec(vec + EV_B2 * H, q);
#ifdef FE_EDGE_T2   // diagnostic lever: two-part split of the first chained layer's operand (forward kernel, fp32 mode)
  if constexpr (!KEEP_D && MODE == GM_X3 && !RM) gemm64_x3_t2(reinterpret_cast<const unsigned *>(img), S.t, S.mp);
  else
#endif
  gemm_e<MODE, 0, RM>(img, S.t, S.mp);
  FE_T(3)   // gemm 1
  if constexpr (FOLD) S.m0 = vsilu2(S.mp);
  else S.m0 = KEEP_D ? vsilu_keep_d(S.mp FE_ACT(a)) : vsilu(S.mp FE_ACT(a));
  if (a.flags & FASTEGNN_F_ATTENTION) {
    S.att = sigmoid_f(vdot(S.m0, vload_vec(vec + EV_ATT * H, q)) + a.attb[0]);
    S.m = vscale(S.m0, S.att);
  } else {
    S.att = 1.f;
    S.m = S.m0;
  }
  FE_T(2)
  S.up = vload_vec(vec + EV_BX1 * H, q);
  gemm_e<MODE, 1, RM>(img, S.m, S.up);
  FE_T(3)
  if constexpr (FOLD) S.u = vsilu2(S.up);
  else S.u = KEEP_D ? vsilu_keep_d(S.up FE_ACT(a)) : vsilu(S.up FE_ACT(a));
  const float sraw = vdot(S.u, vload_vec(vec + EV_WX2 * H, q)) + (a.bx2 ? a.bx2[0] : 0.f);
  S.s = (a.flags & FASTEGNN_F_TANH) ? tanh_f(sraw) : sraw;
  FE_T(4)   // silu 3 + head dot
}
template <bool KEEP_D, int MODE = GM_F32, bool RM = false>
__device__ __forceinline__ void edge_tile_forward(const EdgeArgs &a, const void *img, const float *vec,
                                                  const EdgeIdx &I, int q, EdgeFwdState &S, Vec &pre FE_TP) {
  EdgeRows G;
  edge_gather(a, I, q, G);
  edge_tile_pre<edge_fold_first<MODE>()>(a, vec, I, G, q, S, pre FE_TA);
  edge_tile_mlp<KEEP_D, MODE, RM>(a, img, vec, q, S, pre FE_TA);
}

inline EdgeArgs make_edge_args(const fastegnn_layer_t *L) {
  const float *const *p = L->params;
  const fastegnn_graph_t &g = L->graph;
  EdgeArgs a{L->P, L->QX, L->QX_src ? L->QX_src : L->QX, L->ea_sorted, L->wpack,
             p[FASTEGNN_P_EDGE0_W], p[FASTEGNN_P_EDGE2_B], p[FASTEGNN_P_CR0_B], p[FASTEGNN_P_CR2_W],
             p[FASTEGNN_P_ATT_W], p[FASTEGNN_P_ATT_B], p[FASTEGNN_P_CR2_B], g.rowptr, g.erow, g.col, g.chunk_row, L->aggm, L->aggx,
             g.n_chunks, L->ea, L->flags, L->epsilon};
  a.act_param = L->act_param;
  return a;
}

struct VirtArgs {
  const float *h, *A, *Bc, *x, *vel, *Z, *aggm, *aggx, *svel, *sgrav, *node_attr, *wpack;
  const float *V0W, *c2, *bxv0, *wxv2, *bxx0, *wxx2, *attw, *attb, *N0W, *b3, *b4;
  const int32_t *batch;
  float *h_out, *x_out, *npre, *poolV, *poolX;
  int N, B, C, na, flags;
  float g[3];
  float act_param = 0.f;
};
constexpr int VV_WVR = 0, VV_C2 = 1, VV_BXV0 = 2, VV_WXV2 = 3, VV_BXX0 = 4, VV_WXX2 = 5, VV_ATT = 6, VV_B3 = 7,
              VV_B4 = 8, VV_COUNT = 9;

template <int MODE>
struct VirtFwdState {
  Vec pre, t, vp, v0, v, uxp, uXp;
  typename OperandOf<MODE>::type vs;   // v as the B operand of the two coordinate heads and of the node-MLP block
  float vd[3], vr, att, sx, sX;
};

__device__ __forceinline__ void virt_load_vecs(float *vec, const VirtArgs &a) {
  const int ld = 2 * H + 1 + a.C;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    // the virtual-node parameters are absent (null) for the EGNN baseline (C = 0)
    vec[VV_WVR * H + i] = a.V0W ? a.V0W[(size_t)i * ld + 2 * H] : 0.f;
    vec[VV_C2 * H + i] = a.c2 ? a.c2[i] : 0.f;
    vec[VV_BXV0 * H + i] = a.bxv0 ? a.bxv0[i] : 0.f;
    vec[VV_WXV2 * H + i] = a.wxv2 ? a.wxv2[i] : 0.f;
    vec[VV_BXX0 * H + i] = a.bxx0 ? a.bxx0[i] : 0.f;
    vec[VV_WXX2 * H + i] = a.wxx2 ? a.wxx2[i] : 0.f;
    vec[VV_ATT * H + i] = a.attw ? a.attw[i] : 0.f;
    vec[VV_B3 * H + i] = a.b3 ? a.b3[i] : 0.f;   // node_mlp is absent in the FastRF layer
    vec[VV_B4 * H + i] = a.b4 ? a.b4[i] : 0.f;
  }
}

// forward math of one (16-node tile, channel c); img = resident V2, WXV0, WXX0
template <int MODE = GM_F32>
__device__ __forceinline__ void virt_tile_forward(const VirtArgs &a, const void *img, const float *vec, const Vec &Ai,
                                                  const float xi[3], int b, int c, int q, const float *BcL,
                                                  const float *ZL, VirtFwdState<MODE> &S VF_TP) {
  // BcL / ZL (wave-uniform, LDS): the Bc rows [C][64] and virtual coordinates [3][C] of the graph when the whole tile
  // lies in one graph and the workgroup has staged them; null: read per node from global memory
  const int C = a.C;
  if (ZL) {
    S.vd[0] = ZL[c] - xi[0];
    S.vd[1] = ZL[C + c] - xi[1];
    S.vd[2] = ZL[2 * C + c] - xi[2];
  } else {
    const float *Zb = a.Z + (size_t)b * 3 * C;
    S.vd[0] = Zb[c] - xi[0];
    S.vd[1] = Zb[C + c] - xi[1];
    S.vd[2] = Zb[2 * C + c] - xi[2];
  }
  S.vr = sqrt_f(S.vd[0] * S.vd[0] + S.vd[1] * S.vd[1] + S.vd[2] * S.vd[2]);
#ifdef FE_VIRT_PRE_MFMA   // measured lever, rejected (round 4: virt_fwd 1.336 -> 1.341 ms per step, tools/gpu_lever_virt_pre_mfma.sh)
  if (BcL) {
    // pre = A[n] + Bc[c] + vr * w_vr as ONE rank-2 update on the matrix pipe (as edge_tile_pre): features (vr, 1) of the
    // node against the columns (w_vr | Bc[c]); lanes q = 0 supply w_vr and vr, lanes q = 1 Bc[c] and 1, the others a zero
    // feature.  16 adds + 16 fmas + 8 LDS reads per (tile, channel) become 4 MFMAs + 4 LDS words -- and nothing is gained:
    // this kernel runs two waves per SIMD with its matrix pipe twice as busy as edge_fwd's.
    const int j = lane_id() & 15;
    const float *wsrc = (q == 0 ? vec + VV_WVR * H : BcL + c * H) + j;
    const float f = q == 0 ? S.vr : (q == 1 ? 1.f : 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) S.pre.t[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wsrc[16 * t], f, Ai.t[t], 0, 0, 0);
  } else
#endif
  {
    S.pre = Ai;
    if (BcL) vadd(S.pre, vload_vec(BcL + c * H, q));
    else vadd(S.pre, vload_row(a.Bc + ((size_t)b * C + c) * H, q));
    vaxpy(S.pre, S.vr, vload_vec(vec + VV_WVR * H, q));
  }
  S.t = vsilu(S.pre FE_ACT(a));
  VF_T(1)   // geometry, pre-activation, silu 1
  S.vp = vload_vec(vec + VV_C2 * H, q);
  VF_PRIO_ON();
  gemm_i<MODE>(img, 0, S.t, S.vp);
  VF_PRIO_OFF();
  VF_T(2)   // split + product 1
  S.v0 = vsilu(S.vp FE_ACT(a));
  if (a.flags & FASTEGNN_F_ATTENTION) {
    S.att = sigmoid_f(vdot(S.v0, vload_vec(vec + VV_ATT * H, q)) + a.attb[0]);
    S.v = vscale(S.v0, S.att);
  } else {
    S.att = 1.f;
    S.v = S.v0;
  }
  S.uxp = vload_vec(vec + VV_BXV0 * H, q);
  S.vs = make_operand<MODE>(S.v);   // one split / rounding feeds both coordinate heads and the node MLP
  VF_T(3)   // silu 2 + split
  VF_PRIO_ON();
  gemm_op<MODE>(img, 1, S.vs, S.uxp);
  VF_PRIO_OFF();
  VF_T(4)   // product 2
  float sr = vdot(vsilu(S.uxp FE_ACT(a)), vload_vec(vec + VV_WXV2 * H, q));
  S.sx = (a.flags & FASTEGNN_F_TANH) ? tanh_f(sr) : sr;
  VF_T(5)   // silu + head dot
  S.uXp = vload_vec(vec + VV_BXX0 * H, q);
  VF_PRIO_ON();
  gemm_op<MODE>(img, 2, S.vs, S.uXp);
  VF_PRIO_OFF();
  VF_T(4)
  sr = vdot(vsilu(S.uXp FE_ACT(a)), vload_vec(vec + VV_WXX2 * H, q));
  S.sX = (a.flags & FASTEGNN_F_TANH) ? tanh_f(sr) : sr;
  VF_T(5)
  // (measured in round 3: the two head products and the node-MLP block product issued back to back, activations and
  // dot products after them -- two pipeline drains per channel instead of four -- was slower: 1.71 vs 1.62 ms per step,
  // 42 spilled registers)
}

inline VirtArgs make_virt_args(const fastegnn_layer_t *L) {
  const float *const *p = L->params;
  VirtArgs a{L->h, L->A, L->Bc, L->x, L->vel, L->Z, L->aggm, L->aggx, L->svel, L->sgrav, L->node_attr, L->wpack,
             p[FASTEGNN_P_VIRT0_W], p[FASTEGNN_P_VIRT2_B], p[FASTEGNN_P_CRV0_B], p[FASTEGNN_P_CRV2_W],
             p[FASTEGNN_P_CVV0_B], p[FASTEGNN_P_CVV2_W], p[FASTEGNN_P_ATTV_W], p[FASTEGNN_P_ATTV_B],
             p[FASTEGNN_P_NODE0_W], p[FASTEGNN_P_NODE0_B], p[FASTEGNN_P_NODE2_B], L->batch,
             L->h_out, L->x_out, L->npre, L->poolV, L->poolX, L->N, L->B, L->C, L->na, L->flags,
             {L->gravity[0], L->gravity[1], L->gravity[2]}};
  a.act_param = L->act_param;
  return a;
}
void launch_virt_fwd_pair(const VirtArgs &a, int grid, size_t lds, hipStream_t st);   // virt_fwd_pair.hip
inline size_t virt_lds_bytes(int C, int n_img, int waves = VIRT_WAVES) {
  return (size_t)(n_img * IMG + 16 * H + waves * 16 * TS + C * H + 4 * C) * sizeof(float);
}


}  // namespace fe
