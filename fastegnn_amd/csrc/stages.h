// Device code shared by the forward and backward stage kernels (the backward recomputes the
// forward of a tile instead of storing [E,64] / [N,C,64] activations).
#pragma once
#include "kernels.h"

namespace fe {

constexpr int EDGE_WAVES = 8;
constexpr int VIRT_WAVES = 8;
constexpr int VIRT_BWD_WAVES = 4;   // 1 wave/SIMD: the adjoint of the virtual block needs > 256 registers

struct EdgeArgs {
  const float *P, *QX, *QXs, *ea, *wpack;
  const float *E0W, *b2, *bx1, *wx2, *attw, *attb, *bx2;   // bx2: coordinate-head bias (EGNN baseline) or null
  const int32_t *rowptr, *erow, *col, *chunk_row;
  float *aggm, *aggx;
  int n_chunks, ea_dim, flags;
  float eps;
};

constexpr int EV_WR = 0, EV_WE = 1, EV_B2 = 9, EV_BX1 = 10, EV_WX2 = 11, EV_ATT = 12, EV_COUNT = 13;

__device__ __forceinline__ void edge_load_vecs(float *vec, const EdgeArgs &a) {
  const int ld = 2 * H + 1 + a.ea_dim;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    vec[EV_WR * H + i] = a.E0W[(size_t)i * ld + ((a.flags & FASTEGNN_F_EGNN) ? 0 : 2 * H)];
    for (int k = 0; k < 8; ++k) vec[(EV_WE + k) * H + i] = k < a.ea_dim ? a.E0W[(size_t)i * ld + 2 * H + 1 + k] : 0.f;
    vec[EV_B2 * H + i] = a.b2[i];
    vec[EV_BX1 * H + i] = a.bx1[i];
    vec[EV_WX2 * H + i] = a.wx2[i];
    vec[EV_ATT * H + i] = a.attw ? a.attw[i] : 0.f;
  }
}

struct EdgeFwdState {
  Vec t, mp, m0, m, up, u;
  float d[3], dn[3], r, nrm, att, s, eav[8];
  int row, col;
};

// per-edge indices and scalar attributes; loaded one tile ahead of their use so that only one
// level of the (index -> gathered row) dependent-load chain is exposed per tile
struct EdgeIdx {
  int row, col;
  float eav[8];
};
__device__ __forceinline__ void edge_load_idx(const EdgeArgs &a, int e, EdgeIdx &I) {
  I.row = a.erow[e];
  I.col = a.col[e];
#pragma unroll
  for (int k = 0; k < 8; ++k) I.eav[k] = 0.f;
  if (a.ea_dim > 0) {
    const float *er = a.ea + (size_t)e * a.ea_dim;
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // unconditional (clamped) loads: no branch + wait per attribute
      const float v = er[k < a.ea_dim ? k : 0];
      I.eav[k] = k < a.ea_dim ? v : 0.f;
    }
  }
}

// forward math of one 16-edge tile (shared with the backward kernel for recomputation).
// KEEP_D: pre, S.mp and S.up return silu'(.) of the pre-activations instead of the pre-activations
// (the adjoint needs only the derivatives; one sigmoid serves both).
template <bool KEEP_D>
__device__ __forceinline__ void edge_tile_forward(const EdgeArgs &a, const float *img, const float *vec,
                                                  const EdgeIdx &I, int q, EdgeFwdState &S, Vec &pre FE_TP) {
  S.row = I.row;
  S.col = I.col;
  const float *qrow = a.QXs + (size_t)S.col * QXLD;
  const f32x4 xc = *reinterpret_cast<const f32x4 *>(qrow + H);
  const f32x4 xr = *reinterpret_cast<const f32x4 *>(a.QX + (size_t)S.row * QXLD + H);
  S.d[0] = xr[0] - xc[0];
  S.d[1] = xr[1] - xc[1];
  S.d[2] = xr[2] - xc[2];
  S.r = S.d[0] * S.d[0] + S.d[1] * S.d[1] + S.d[2] * S.d[2];
  S.nrm = sqrt_f(S.r);
  if (a.flags & FASTEGNN_F_NORMALIZE) {
    const float inv = rcp_f(S.nrm + a.eps);
    S.dn[0] = S.d[0] * inv; S.dn[1] = S.d[1] * inv; S.dn[2] = S.d[2] * inv;
  } else {
    S.dn[0] = S.d[0]; S.dn[1] = S.d[1]; S.dn[2] = S.d[2];
  }
  FE_T(0)   // indices + coordinates arrived
  pre = vload_row(a.P + (size_t)S.row * H, q);
  vadd(pre, vload_row(qrow, q));
  vaxpy(pre, S.r, vload_vec(vec + EV_WR * H, q));
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    S.eav[k] = I.eav[k];
    if (k < a.ea_dim) vaxpy(pre, S.eav[k], vload_vec(vec + (EV_WE + k) * H, q));
  }
  FE_T(1)   // gathered rows arrived, pre-activation formed
  S.t = KEEP_D ? vsilu_keep_d(pre) : vsilu(pre);
  FE_T(2)   // silu 1
  S.mp = vload_vec(vec + EV_B2 * H, q);
  gemm64(img + 0 * IMG, S.t, S.mp);
  FE_T(3)   // gemm 1
  S.m0 = KEEP_D ? vsilu_keep_d(S.mp) : vsilu(S.mp);
  if (a.flags & FASTEGNN_F_ATTENTION) {
    S.att = sigmoid_f(vdot(S.m0, vload_vec(vec + EV_ATT * H, q)) + a.attb[0]);
    S.m = vscale(S.m0, S.att);
  } else {
    S.att = 1.f;
    S.m = S.m0;
  }
  FE_T(2)
  S.up = vload_vec(vec + EV_BX1 * H, q);
  gemm64(img + 1 * IMG, S.m, S.up);
  FE_T(3)
  S.u = KEEP_D ? vsilu_keep_d(S.up) : vsilu(S.up);
  const float sraw = vdot(S.u, vload_vec(vec + EV_WX2 * H, q)) + (a.bx2 ? a.bx2[0] : 0.f);
  S.s = (a.flags & FASTEGNN_F_TANH) ? tanh_f(sraw) : sraw;
  FE_T(4)   // silu 3 + head dot
}

inline EdgeArgs make_edge_args(const fastegnn_layer_t *L) {
  const float *const *p = L->params;
  const fastegnn_graph_t &g = L->graph;
  EdgeArgs a{L->P, L->QX, L->QX_src ? L->QX_src : L->QX, L->ea_sorted, L->wpack,
             p[FASTEGNN_P_EDGE0_W], p[FASTEGNN_P_EDGE2_B], p[FASTEGNN_P_CR0_B], p[FASTEGNN_P_CR2_W],
             p[FASTEGNN_P_ATT_W], p[FASTEGNN_P_ATT_B], p[FASTEGNN_P_CR2_B], g.rowptr, g.erow, g.col, g.chunk_row, L->aggm, L->aggx,
             g.n_chunks, L->ea, L->flags, L->epsilon};
  return a;
}

struct VirtArgs {
  const float *h, *A, *Bc, *x, *vel, *Z, *aggm, *aggx, *svel, *sgrav, *node_attr, *wpack;
  const float *V0W, *c2, *bxv0, *wxv2, *bxx0, *wxx2, *attw, *attb, *N0W, *b3, *b4;
  const int32_t *batch;
  float *h_out, *x_out, *npre, *poolV, *poolX;
  int N, B, C, na, flags;
  float g[3];
};
constexpr int VV_WVR = 0, VV_C2 = 1, VV_BXV0 = 2, VV_WXV2 = 3, VV_BXX0 = 4, VV_WXX2 = 5, VV_ATT = 6, VV_B3 = 7,
              VV_B4 = 8, VV_COUNT = 9;

struct VirtFwdState {
  Vec pre, t, vp, v0, v, uxp, uXp;
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
    vec[VV_B3 * H + i] = a.b3[i];
    vec[VV_B4 * H + i] = a.b4[i];
  }
}

// forward math of one (16-node tile, channel c); img = resident V2, WXV0, WXX0
__device__ __forceinline__ void virt_tile_forward(const VirtArgs &a, const float *img, const float *vec, const Vec &Ai,
                                                  const float xi[3], int b, int c, int q, VirtFwdState &S) {
  const int C = a.C;
  const float *Zb = a.Z + (size_t)b * 3 * C;
  S.vd[0] = Zb[c] - xi[0];
  S.vd[1] = Zb[C + c] - xi[1];
  S.vd[2] = Zb[2 * C + c] - xi[2];
  S.vr = sqrt_f(S.vd[0] * S.vd[0] + S.vd[1] * S.vd[1] + S.vd[2] * S.vd[2]);
  S.pre = Ai;
  vadd(S.pre, vload_row(a.Bc + ((size_t)b * C + c) * H, q));
  vaxpy(S.pre, S.vr, vload_vec(vec + VV_WVR * H, q));
  S.t = vsilu(S.pre);
  S.vp = vload_vec(vec + VV_C2 * H, q);
  gemm64(img + 0 * IMG, S.t, S.vp);
  S.v0 = vsilu(S.vp);
  if (a.flags & FASTEGNN_F_ATTENTION) {
    S.att = sigmoid_f(vdot(S.v0, vload_vec(vec + VV_ATT * H, q)) + a.attb[0]);
    S.v = vscale(S.v0, S.att);
  } else {
    S.att = 1.f;
    S.v = S.v0;
  }
  S.uxp = vload_vec(vec + VV_BXV0 * H, q);
  gemm64(img + 1 * IMG, S.v, S.uxp);
  float sr = vdot(vsilu(S.uxp), vload_vec(vec + VV_WXV2 * H, q));
  S.sx = (a.flags & FASTEGNN_F_TANH) ? tanh_f(sr) : sr;
  S.uXp = vload_vec(vec + VV_BXX0 * H, q);
  gemm64(img + 2 * IMG, S.v, S.uXp);
  sr = vdot(vsilu(S.uXp), vload_vec(vec + VV_WXX2 * H, q));
  S.sX = (a.flags & FASTEGNN_F_TANH) ? tanh_f(sr) : sr;
}

inline VirtArgs make_virt_args(const fastegnn_layer_t *L) {
  const float *const *p = L->params;
  VirtArgs a{L->h, L->A, L->Bc, L->x, L->vel, L->Z, L->aggm, L->aggx, L->svel, L->sgrav, L->node_attr, L->wpack,
             p[FASTEGNN_P_VIRT0_W], p[FASTEGNN_P_VIRT2_B], p[FASTEGNN_P_CRV0_B], p[FASTEGNN_P_CRV2_W],
             p[FASTEGNN_P_CVV0_B], p[FASTEGNN_P_CVV2_W], p[FASTEGNN_P_ATTV_W], p[FASTEGNN_P_ATTV_B],
             p[FASTEGNN_P_NODE0_W], p[FASTEGNN_P_NODE0_B], p[FASTEGNN_P_NODE2_B], L->batch,
             L->h_out, L->x_out, L->npre, L->poolV, L->poolX, L->N, L->B, L->C, L->na, L->flags,
             {L->gravity[0], L->gravity[1], L->gravity[2]}};
  return a;
}
inline size_t virt_lds_bytes(int C, int n_img, int waves = VIRT_WAVES) {
  return (size_t)(n_img * IMG + 16 * H + waves * 16 * TS + C * H + 4 * C) * sizeof(float);
}


}  // namespace fe
