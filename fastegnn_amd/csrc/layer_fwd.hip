// Forward stages S1..S5 of one E_GCL_vel layer (reference: models/FastEGNN.py:192-223).
// Math per stage: oracle/factored.py (same stage names); layout conventions: common.h.
#include <cstdlib>
#include "stages.h"

namespace fe {

// =====================================================================================
// S1 node_pre:  P = h W1a^T + b1, Q = h W1b^T, A = h V1a^T  (first-Linear factorisation of
// edge_mlp.0 / edge_mlp_virtual.0, FastEGNN.py:103,114), velocity / gravity heads (:139-142).
// =====================================================================================
struct NodePreArgs {
  const float *h, *x, *wpack;
  const float *b1, *bv0, *wv2, *bv2, *bg0, *wg2, *bg2;
  float *P, *QX, *A, *svel, *sgrav;
  int N, gravity, has_vel;
  const float *vel, *wv0;   // FastRF: velocity scale from ||vel|| through coord_mlp_vel.0.weight [H,1]
  int C;
  int flags = 0;
  float act_param = 0.f;
};

// MODE: GM_X3 (bf16x3 products of the split images: fp32-grade, 2.7 x fewer matrix cycles than the fp32-input MFMA these
// node-level kernels ran on through round 2 -- 2048 -> 768 cycles per 64x64 product and tile, + one operand split per
// input) or GM_BF16 (bf16 operand mode: one product of the RNE-rounded activation)
template <int MODE>
__global__ __launch_bounds__(64 * NODE_PRE_WAVES) void node_pre_fwd_kernel(NodePreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned *img = reinterpret_cast<unsigned *>(lds);   // 5 split images: W1A W1B V1A WVEL0 WG0
  float *vec = lds + 5 * IMG3;                         // b1 bv0 wv2 bg0 wg2 wv0
  const int nimg = a.gravity ? 5 : 4;
  load_images_x3(img, wpack_x3(a.wpack, a.C, I_W1A), nimg);
  // (P in units of ln 2 when the edge stage's first layer is folded: the W1a / W1b images carry log2(e), pack.hip)
  if constexpr (edge_fold_first<MODE>()) { for (int i = threadIdx.x; i < H; i += blockDim.x) vec[i] = a.b1 ? a.b1[i] * LOG2E_F : 0.f; }
  else load_floats(vec + 0 * H, a.b1, H);
  load_floats(vec + 1 * H, a.bv0, H);
  load_floats(vec + 2 * H, a.wv2, H);
  load_floats(vec + 3 * H, a.bg0, H);
  load_floats(vec + 4 * H, a.wg2, H);
  load_floats(vec + 5 * H, a.wv0, H);
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4;
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ntiles = (a.N + 15) >> 4;
  const float bv2 = (a.has_vel || a.wv0) ? a.bv2[0] : 0.f;
  const float bg2 = a.gravity ? a.bg2[0] : 0.f;
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int n = tile * 16 + j;
    const bool valid = n < a.N;
    const int nc = valid ? n : a.N - 1;
    const typename OperandOf<MODE>::type hv = make_operand<MODE>(vload_row(a.h + (size_t)nc * H, q));   // one split / rounding feeds all products
    Vec acc = vload_vec(vec, q);
    gemm_op<MODE>(img, 0, hv, acc);
    if (valid) vstore_row(a.P + (size_t)n * H, q, acc);
    acc = vzero();
    gemm_op<MODE>(img, 1, hv, acc);
    if (valid) {
      vstore_row(a.QX + (size_t)n * QXLD, q, acc);
      if (q == 0) {
        f32x4 xv = {a.x[(size_t)n * 3], a.x[(size_t)n * 3 + 1], a.x[(size_t)n * 3 + 2], 0.f};
        *reinterpret_cast<f32x4 *>(a.QX + (size_t)n * QXLD + H) = xv;
      }
    }
    acc = vzero();
    gemm_op<MODE>(img, 2, hv, acc);
    if (valid) vstore_row(a.A + (size_t)n * H, q, acc);
    float s = 0.f;
    if (a.has_vel) {
      acc = vload_vec(vec + 1 * H, q);
      gemm_op<MODE>(img, 3, hv, acc);
      s = vdot(vsilu(acc FE_ACT(a)), vload_vec(vec + 2 * H, q)) + bv2;
    } else if (a.wv0) {   // FastRF.py:139: coord_mlp_vel(||vel||), the norm is detached (:169)
      const float vx = a.vel[(size_t)nc * 3], vy = a.vel[(size_t)nc * 3 + 1], vz = a.vel[(size_t)nc * 3 + 2];
      acc = vload_vec(vec + 1 * H, q);
      vaxpy(acc, sqrt_f(vx * vx + vy * vy + vz * vz), vload_vec(vec + 5 * H, q));
      s = vdot(vsilu(acc FE_ACT(a)), vload_vec(vec + 2 * H, q)) + bv2;
    }
    if (valid && q == 0) a.svel[n] = s;
    if (a.gravity) {
      acc = vload_vec(vec + 3 * H, q);
      gemm_op<MODE>(img, 4, hv, acc);
      s = vdot(vsilu(acc FE_ACT(a)), vload_vec(vec + 4 * H, q)) + bg2;
      if (valid && q == 0) a.sgrav[n] = s;
    }
  }
}

int node_pre_forward(const fastegnn_layer_t *L, hipStream_t st) {
  FE_REQUIRE(L->h && L->x && L->wpack && L->P && L->QX && L->A && L->svel, "node_pre_forward: null buffer");
  if (L->N == 0) return FASTEGNN_OK;
  const bool grav = has(L, FASTEGNN_F_GRAVITY);
  FE_REQUIRE(!grav || L->sgrav, "node_pre_forward: sgrav null");
  const float *const *p = L->params;
  NodePreArgs a{L->h, L->x, L->wpack, p[FASTEGNN_P_EDGE0_B], p[FASTEGNN_P_VEL0_B], p[FASTEGNN_P_VEL2_W],
                p[FASTEGNN_P_VEL2_B], p[FASTEGNN_P_GRAV0_B], p[FASTEGNN_P_GRAV2_W], p[FASTEGNN_P_GRAV2_B],
                L->P, L->QX, L->A, L->svel, L->sgrav, L->N, grav ? 1 : 0,
                (p[FASTEGNN_P_VEL0_W] && !has(L, FASTEGNN_F_RF)) ? 1 : 0,
                L->vel, has(L, FASTEGNN_F_RF) ? p[FASTEGNN_P_VEL0_W] : nullptr, L->C, L->flags, L->act_param};
  FE_REQUIRE(!has(L, FASTEGNN_F_RF) || (L->vel && p[FASTEGNN_P_VEL0_W] && p[FASTEGNN_P_VEL0_B] && p[FASTEGNN_P_VEL2_W] &&
                                         p[FASTEGNN_P_VEL2_B]),
             "node_pre_forward: FastRF needs vel and the coord_mlp_vel parameters");
  const int ntiles = (L->N + 15) / 16;
  int grid = cdiv(ntiles, NODE_PRE_WAVES);
  if (grid > 256) grid = 256;
  const size_t lds = (5 * IMG3 + 6 * H) * sizeof(float);
  {
    ProfScope _ps_node_pre_fwd_kernel(K_NODE_PRE_FWD, st);
    if (has(L, FASTEGNN_F_BF16)) hipLaunchKernelGGL(node_pre_fwd_kernel<GM_BF16>, dim3(grid), dim3(64 * NODE_PRE_WAVES), lds, st, a);
    else hipLaunchKernelGGL(node_pre_fwd_kernel<GM_NODE_PRE_FWD>, dim3(grid), dim3(64 * NODE_PRE_WAVES), lds, st, a);
  }
  return check_launch("node_pre_fwd_kernel");
}

// =====================================================================================
// S2a graph_xsum: per-graph sum of coordinates and node count (global_mean_pool(coord), :212)
// =====================================================================================
// A wave owns XSUM_PER_WAVE consecutive nodes and walks them 64 at a time (lane = node).  (Round 4 measured 256 nodes per wave --
// four iterations instead of sixteen, four times the waves: 26 us per launch at 100 000 nodes either way; with one big graph the
// launch is bound by the same-address atomics of its 98 - 391 waves, not by the loop.)  data_batch is ascending, so a
// 64-node group normally lies inside one graph: the lanes keep partial sums and the wave leaves ONE atomic set per
// (wave, graph) run; groups that straddle graphs (mini-batches of small graphs) are reduced by a segmented scan over the
// lanes (data_batch is sorted, so a graph is a run of lanes) and add once per (group, graph).
#ifndef FE_XSUM_PER_WAVE
#define FE_XSUM_PER_WAVE 1024
#endif
constexpr int XSUM_PER_WAVE = FE_XSUM_PER_WAVE;
// (round 6: the nodes per wave are a launch argument -- XSUM_PER_WAVE for large inputs; a small input (a 12 500-node shard, the N-body
//  mini-batches) gets shorter walks on more waves: at 1 024 nodes per wave a shard's launch was 13 waves x 16 dependent iterations = 15 us)
__global__ __launch_bounds__(256) void graph_xsum_kernel(const float *x, const int32_t *batch, int N, float *xsum, int PER_WAVE) {
  const int l = lane_id();
  const int n0 = global_wave_id() * PER_WAVE, n1 = min(N, n0 + PER_WAVE);
  if (n0 >= n1) return;
  int cur = -1;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  // (ONE atomic instruction per flush: lanes 0..3 add the four components of the graph's row -- four single-lane atomics per wave
  // made the launch 26 us long at 100 000 nodes in one graph, every wave queueing on the same cache line four times)
  auto flush = [&]() {
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = s[k];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (l == k) mine = v;
      s[k] = 0.f;
    }
    if (l < 4) atomicAdd(&xsum[cur * 4 + l], mine);
  };
  for (int base = n0; base < n1; base += 64) {
    const int n = base + l;
    const bool ok = n < n1;
    const int b = batch[ok ? n : n1 - 1];
    const int bf = __builtin_amdgcn_readfirstlane(b), bl = batch[min(base + 63, n1 - 1)];
    float xv[3] = {0.f, 0.f, 0.f};
    if (ok) { xv[0] = x[(size_t)n * 3]; xv[1] = x[(size_t)n * 3 + 1]; xv[2] = x[(size_t)n * 3 + 2]; }
    if (bf == bl) {   // (wave-uniform) the whole group lies in graph bf
      if (bf != cur) {
        if (cur >= 0) flush();
        cur = bf;
      }
      s[0] += xv[0]; s[1] += xv[1]; s[2] += xv[2]; s[3] += ok ? 1.f : 0.f;
    } else {
      if (cur >= 0) flush();
      cur = -1;
      // segmented inclusive scan: after it the LAST lane of every run of equal graph ids holds the run's sums
      const int bb = ok ? b : -1;
      float v[4] = {xv[0], xv[1], xv[2], ok ? 1.f : 0.f};
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int bu = __shfl_up(bb, off);
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = __shfl_up(v[k], off);
        if (l >= off && bu == bb) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] += u[k];
        }
      }
      const int bn = __shfl_down(bb, 1);
      if (ok && (l == 63 || bn != bb)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&xsum[b * 4 + k], v[k]);
      }
    }
  }
  if (cur >= 0) flush();
}

int graph_xsum(const fastegnn_layer_t *L, hipStream_t st) {
  FE_REQUIRE(L->xsum && L->batch && L->x, "graph_xsum: null buffer");
  (void)hipMemsetAsync(L->xsum, 0, (size_t)L->B * 4 * sizeof(float), st);
  // nodes per wave: ~64 waves in flight for a small input (same-address atomics: one set per wave), XSUM_PER_WAVE at most
  int per_wave = (cdiv(L->N, 64) + 63) / 64 * 64;
  per_wave = per_wave < 64 ? 64 : (per_wave > XSUM_PER_WAVE ? XSUM_PER_WAVE : per_wave);
  if (L->N > 0) { ProfScope _ps_graph_xsum_kernel(K_XSUM, st); hipLaunchKernelGGL(graph_xsum_kernel, dim3(cdiv(L->N, 4 * per_wave)), dim3(256), 0, st, L->x, L->batch, L->N, L->xsum, per_wave); }   // 4 waves x per_wave nodes
  return check_launch("graph_xsum_kernel");
}

// =====================================================================================
// S2b graph_pre: centroid, Gram matrix m_X (:212-214) and the per-(graph,channel) part of
// edge_mlp_virtual.0:  Bc[b,c,:] = V1b Hv[b,:,c] + V1d mX[b][:,c] + c1   (:114)
// =====================================================================================
struct GraphPreArgs {
  const float *xsum, *Z, *HvT, *V0W, *V0B;
  float *Bc;
  int B, C, bf16;
};
// grid (B, GP_SPLIT): the C*64 outputs of a graph are dealt to GP_SPLIT workgroups (each recomputes the graph's small
// Gram matrix): with one big graph the stage was a single workgroup on one CU
constexpr int GP_SPLIT = 4;
__global__ __launch_bounds__(256) void graph_pre_fwd_kernel(GraphPreArgs a) {
  extern __shared__ float sm[];
  const int C = a.C, b = blockIdx.x, ld = 2 * H + 1 + C;
  float *mz = sm;           // [3][C]
  float *mX = sm + 3 * C;   // [C][C]
  const float cnt = fmaxf(a.xsum[b * 4 + 3], 1.f);
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    int k = i / C;
    mz[i] = a.Z[(size_t)b * 3 * C + i] - a.xsum[b * 4 + k] / cnt;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256) {
    int c = i / C, d = i % C;
    mX[i] = mz[c] * mz[d] + mz[C + c] * mz[C + d] + mz[2 * C + c] * mz[2 * C + d];
  }
  __syncthreads();
  for (int i = blockIdx.y * 256 + threadIdx.x; i < C * H; i += 256 * gridDim.y) {
    int c = i >> 6, o = i & 63;
    const float *w = a.V0W + (size_t)o * ld;
    const float *hv = a.HvT + ((size_t)b * C + c) * H;
    float acc = a.V0B[o];
    if (a.bf16) { for (int k = 0; k < H; ++k) acc += round_bf(w[H + k]) * round_bf(hv[k]); }   // V1b Hv: bf16 operands
    else { for (int k = 0; k < H; ++k) acc += w[H + k] * hv[k]; }
    for (int d = 0; d < C; ++d) acc += w[2 * H + 1 + d] * mX[d * C + c];   // Gram columns: always fp32
    a.Bc[((size_t)b * C + c) * H + o] = acc;
  }
}
int graph_pre_forward(const fastegnn_layer_t *L, hipStream_t st) {
  FE_REQUIRE(L->xsum && L->Z && L->HvT && L->Bc, "graph_pre_forward: null buffer");
  GraphPreArgs a{L->xsum, L->Z, L->HvT, L->params[FASTEGNN_P_VIRT0_W], L->params[FASTEGNN_P_VIRT0_B], L->Bc, L->B, L->C,
                 has(L, FASTEGNN_F_BF16) ? 1 : 0};
  const size_t lds = (size_t)(3 * L->C + L->C * L->C) * sizeof(float);
  { ProfScope _ps_graph_pre_fwd_kernel(K_GRAPH_PRE_FWD, st); hipLaunchKernelGGL(graph_pre_fwd_kernel, dim3(L->B, GP_SPLIT), dim3(256), lds, st, a); }
  return check_launch("graph_pre_fwd_kernel");
}

// =====================================================================================
// S3 edge: coord2radial (:180-189) + edge_model (:102-108) + real part of coord_model_vel
// (:125-133) + the segment means of node_model / coord_model_vel (:155, :287-294), fused.
// One wave walks an edge-balanced chunk of CSR rows in 16-edge tiles; sums stay in registers
// until the row changes, so every row is written exactly once (no atomics, deterministic).
// =====================================================================================
constexpr int EDGE_FWD_IMG_FLOATS = 2 * IMG3;
template <int MODE>
__global__ __launch_bounds__(64 * EDGE_FWD_WAVES) void edge_fwd_kernel(EdgeArgs a, int C) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef FE_ISA_CONST   // assembly-only builds of tools/isa_budget.py: the layer flags as a constant (straight-line code of ONE configuration)
  a.flags = FE_ISA_CONST; a.ea_dim = 2; a.bx2 = nullptr; a.attw = nullptr;
#endif
  float *img = lds;                              // W2, WX1 (fp32 or split images)
  float *vec = lds + EDGE_FWD_IMG_FLOATS;        // EV_COUNT vectors
  float *tiles = vec + EV_COUNT * H;             // per wave: [16][TS] + [16][4]
  load_images_x3(reinterpret_cast<unsigned *>(img), wpack_x3(a.wpack, C, I_W2), 2);
  constexpr bool FOLD = edge_fold<false, MODE, false>();   // S.m arrives as log2(e) x the message: 1/deg absorbs the factor
  edge_load_vecs(vec, a, FOLD);
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4, wv = wave_id();
  float *mt = tiles + wv * (16 * TS + 64);
  float *xt = mt + 16 * TS;
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool mean = !(a.flags & FASTEGNN_F_COORDS_SUM);
  FE_T0()
  // this wave's share: a contiguous run of whole rows holding ~E/nwaves edges (chunk_row marks the row
  // boundary nearest to every CHUNK_EDGES-th edge)
  const int c0 = (int)((long)wave * a.n_chunks / nwaves), c1 = (int)((long)(wave + 1) * a.n_chunks / nwaves);
  {
    const int r0 = a.chunk_row[c0], r1 = a.chunk_row[c1];
    if (r0 < r1) {
      const int e0 = a.rowptr[r0], e1 = a.rowptr[r1];
      int cur = -1;
      float acc = 0.f, accx = 0.f;
      FE_T(7)   // chunk bookkeeping
      int cnt = 0;   // edges of the current row seen so far == its in-degree at flush time (rows never straddle waves)
      auto flush = [&]() {
        const float inv = rcp_f((float)cnt);
        a.aggm[(size_t)cur * H + l] = acc * (FOLD ? inv * LN2_F : inv);
        if (l < 3) a.aggx[(size_t)cur * 3 + l] = mean ? accx * inv : accx;
      };
      // rows without edges inside this wave's range get their zeros here (no memset of aggm / aggx ahead of the kernel):
      // the wave owns the rows [r0, r1) and writes each of them exactly once
      auto zero_rows = [&](int ra, int rb) {
        for (int r = ra; r < rb; ++r) {
          a.aggm[(size_t)r * H + l] = 0.f;
          if (l < 3) a.aggx[(size_t)r * 3 + l] = 0.f;
        }
      };
      EdgeIdx cur_i, nxt_i;
      if (e0 < e1) edge_load_idx(a, min(e0 + j, e1 - 1), cur_i);
      for (int base = e0; base < e1; base += 16) {
        const int nvalid = min(16, e1 - base);
        nxt_i = cur_i;
        if (base + 16 < e1) edge_load_idx(a, min(base + 16 + j, e1 - 1), nxt_i);   // next tile's indices in flight
        EdgeFwdState S;
        Vec pre;
        edge_tile_forward<false, MODE>(a, img, vec, cur_i, q, S, pre FE_TA);
        cur_i = nxt_i;
        tile_store(mt, j, q, S.m);
        if (q == 0) {
          xt[j * 4 + 0] = S.dn[0] * S.s;
          xt[j * 4 + 1] = S.dn[1] * S.s;
          xt[j * 4 + 2] = S.dn[2] * S.s;
        }
        __builtin_amdgcn_wave_barrier();
        FE_T(5)   // transpose tile to LDS
        // hidden-on-lane column of the tile: all 16 LDS reads issued up front, the row-boundary walk
        // below then runs on registers and scalar compares only
        float mv[16], xv[16];
#pragma unroll
        for (int ee = 0; ee < 16; ++ee) {
          mv[ee] = mt[ee * TS + l];
          xv[ee] = xt[ee * 4 + (l & 3)];
        }
        const int rowv = S.row;
        // where the row changes inside the tile: ONE lane compare with the lane below (DPP) and a ballot instead of sixteen
        // v_readlane + scalar compares; the row id itself is read only at a change (~2 per tile at degree 19)
#ifndef FE_WALK_READLANE
        const int prevrow = __builtin_amdgcn_update_dpp(rowv, rowv, 0x111, 0xf, 0xf, false);   // row_shr:1 (lane 0 of a row keeps its own)
        const unsigned long long chg = __builtin_amdgcn_ballot_w64(j == 0 ? rowv != cur : rowv != prevrow);
        const unsigned starts = (unsigned)chg & 0xffffu;      // lanes 0..15 (q = 0): one bit per edge of the tile
#endif
#pragma unroll
        for (int ee = 0; ee < 16; ++ee) {
          if (ee < nvalid) {
#ifndef FE_WALK_READLANE
            if ((starts >> ee) & 1u) {
              const int rw = __builtin_amdgcn_readlane(rowv, ee);
#else
            const int rw = __builtin_amdgcn_readlane(rowv, ee);
            if (rw != cur) {
#endif
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
        __builtin_amdgcn_wave_barrier();
        FE_T(6)   // row-segmented reduction
      }
      if (cur >= 0) flush();
      zero_rows(cur >= 0 ? cur + 1 : r0, r1);
    }
  }
  FE_TEND()
}

int edge_forward(const fastegnn_layer_t *L, hipStream_t st) {
  FE_REQUIRE(L->P && L->QX && L->aggm && L->aggx && L->wpack, "edge_forward: null buffer");
  FE_REQUIRE(L->ea <= 7, "edge_forward: edge_attr_nf > 7 unsupported");
  FE_REQUIRE(!has(L, FASTEGNN_F_ATTENTION) || (L->params[FASTEGNN_P_ATT_W] && L->params[FASTEGNN_P_ATT_B]),
             "edge_forward: attention params null");
  const fastegnn_graph_t &g = L->graph;
  if (g.n_edges == 0 || L->N == 0) {   // nothing to walk: the aggregates are zero (with edges the kernel writes every row)
    (void)hipMemsetAsync(L->aggm, 0, (size_t)L->N * H * sizeof(float), st);
    (void)hipMemsetAsync(L->aggx, 0, (size_t)L->N * 3 * sizeof(float), st);
    return check_launch("edge_forward(memset)");
  }
  FE_REQUIRE(g.rowptr && g.erow && g.col && g.chunk_row && (L->ea == 0 || L->ea_sorted), "edge_forward: null graph");
  EdgeArgs a = make_edge_args(L);
  FE_REQUIRE((size_t)L->N * QXLD < (1u << 30) && (size_t)g.n_src * QXLD < (1u << 30) && (size_t)g.n_edges * 8 < (1u << 30),
             "edge_forward: tables exceed the 32-bit offset range of the gather path");
#if FE_EDGE_FWD32   // the 32-edge lever kernel (edge_fwd32.hip) is only part of a `make lever32` build
  if (edge_forward32_applies(L)) return edge_forward32(L, st);
#endif
  // one workgroup per CU once there are >= 256 x 16 row chunks; small graphs spread their chunks (32 edges) over as
  // many waves as there are chunks instead of serialising them in a few workgroups (the N-body mini-batches)
  int grid = cdiv(g.n_chunks, EDGE_FWD_WAVES);
  if (grid > 256) grid = 256;
  const size_t lds = (EDGE_FWD_IMG_FLOATS + EV_COUNT * H + EDGE_FWD_WAVES * (16 * TS + 64)) * sizeof(float);
  {
    ProfScope _ps_edge_fwd_kernel(K_EDGE_FWD, st);
    if (has(L, FASTEGNN_F_BF16)) hipLaunchKernelGGL(edge_fwd_kernel<GM_BF16>, dim3(grid), dim3(64 * EDGE_FWD_WAVES), lds, st, a, L->C);
    else hipLaunchKernelGGL(edge_fwd_kernel<GM_EDGE_FWD>, dim3(grid), dim3(64 * EDGE_FWD_WAVES), lds, st, a, L->C);
  }
  return check_launch("edge_fwd_kernel");
}

// =====================================================================================
// S4 virt: edge_mode_virtual (:111-119), virtual part of coord_model_vel (:136-142),
// coord_model_virtual (:146-150), node_model (:153-166) and the pools of node_model_virtual
// (:170), fused.  A wave owns 16 nodes and loops over the C virtual channels; the K = H*C
// contraction of node_mlp.0 accumulates in MFMA registers, v [N,C,H] never reaches HBM.
// =====================================================================================

// LDS: images V2, WXV0, WXX0 | two W3c[c] stage slots | vectors | pools | Bc / Z rows of the graph in flight
}  // namespace fe
#include "virt_fwd.h"   // VIRT_FWD_IMG_FLOATS, virt_fwd_lds_bytes, virt_fwd_kernel<MODE, PAIR>
namespace fe {

int virt_forward(const fastegnn_layer_t *L, hipStream_t st) {
  const bool egnn = has(L, FASTEGNN_F_EGNN);
  FE_REQUIRE(L->h && L->A && L->x && L->vel && L->aggm && L->aggx && L->svel && L->npre && L->h_out && L->x_out &&
                 L->batch && L->wpack,
             "virt_forward: null buffer");
  FE_REQUIRE(egnn ? L->C == 0 : (L->Bc && L->Z && L->poolV && L->poolX && L->C >= 1 && L->C <= 64),
             "virt_forward: virtual_channels must be in [1,64] (0 with FASTEGNN_F_EGNN) and the virtual buffers non-null");
  FE_REQUIRE(L->na == 0 || L->node_attr, "virt_forward: node_attr null");
  if (L->C > 0) {
    (void)hipMemsetAsync(L->poolV, 0, (size_t)L->B * L->C * H * sizeof(float), st);
    (void)hipMemsetAsync(L->poolX, 0, (size_t)L->B * 3 * L->C * sizeof(float), st);
  }
  if (L->N == 0) return check_launch("virt_forward(memset)");
  VirtArgs a = make_virt_args(L);
  // two waves per tile (PAIR) while that puts every tile of the input into ONE step of some workgroup: N <= 256 x 4 tiles
  static const bool pair_off = getenv("FASTEGNN_VIRT_FWD_PAIR") && atoi(getenv("FASTEGNN_VIRT_FWD_PAIR")) == 0;   // (A/B switch)
  const bool pair = GM_VIRT_FWD == GM_F16 && !pair_off && !has(L, FASTEGNN_F_BF16) && !has(L, FASTEGNN_F_RF) && !egnn && L->C >= 2 &&
                    cdiv(L->N, 16 * (VIRT_WAVES / 2)) <= 256 && virt_fwd_lds_bytes(L->C, true) <= 160 * 1024;
  const int ntg = cdiv(L->N, 16 * (pair ? VIRT_WAVES / 2 : VIRT_WAVES));
  int grid = ntg < 256 ? ntg : 256;   // one workgroup per CU (LDS), each with an equal share of the tiles
  {
    ProfScope _ps_virt_fwd_kernel(K_VIRT_FWD, st);
    const size_t lds = virt_fwd_lds_bytes(L->C, pair);
    if (has(L, FASTEGNN_F_BF16)) hipLaunchKernelGGL(virt_fwd_kernel<GM_BF16>, dim3(grid), dim3(64 * VIRT_WAVES), lds, st, a);
    else if (pair) launch_virt_fwd_pair(a, grid, lds, st);   // (virt_fwd_pair.hip: its own translation unit, see virt_fwd.h)
    else hipLaunchKernelGGL(virt_fwd_kernel<GM_VIRT_FWD>, dim3(grid), dim3(64 * VIRT_WAVES), lds, st, a);
  }
  return check_launch("virt_fwd_kernel");
}

// =====================================================================================
// S5 graph_post: coord_model_virtual's mean + residual (:148-149) and node_model_virtual
// (:168-177) on the B*C (graph, channel) rows.
// =====================================================================================
struct GraphPostArgs {
  const float *xsum, *Z, *HvT, *poolV, *poolX, *wpack, *b5, *b6;
  float *Z_out, *HvT_out;
  int B, C, flags;
  float act_param = 0.f;
};
__global__ __launch_bounds__(256) void graph_post_fwd_kernel(GraphPostArgs a) {
  const bool bf = a.flags & FASTEGNN_F_BF16;   // bf16 operand mode: activations rounded, images hold rounded weights
  // (the three images are read straight from global memory: with B*C = 16 rows the stage is ONE tile, and staging 48 KB
  // through LDS first measured slower -- 14.1 vs 13.3 us per launch)
  const int l = lane_id(), j = l & 15, q = l >> 4;
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  const int M = a.B * a.C, ntiles = (M + 15) >> 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.B * 3 * a.C; i += gridDim.x * blockDim.x) {
    const int b = i / (3 * a.C);
    a.Z_out[i] = a.Z[i] + a.poolX[i] / fmaxf(a.xsum[b * 4 + 3], 1.f);
  }
  if (a.flags & FASTEGNN_F_RF) {   // FastRF.py:186: the virtual node features pass through
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M * H; i += gridDim.x * blockDim.x) a.HvT_out[i] = a.HvT[i];
    return;
  }
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int m = tile * 16 + j;
    const bool valid = m < M;
    const int mc = valid ? m : M - 1;
    const int b = mc / a.C;
    const float inv = 1.0f / fmaxf(a.xsum[b * 4 + 3], 1.f);
    const Vec hv = vload_row(a.HvT + (size_t)mc * H, q);
    const Vec pm = vscale(vload_row(a.poolV + (size_t)mc * H, q), inv);
    Vec z5 = vload_vec(a.b5, q);
    gemm64(a.wpack + (size_t)I_W5A * IMG, bf ? vround(hv) : hv, z5);
    gemm64(a.wpack + (size_t)I_W5B * IMG, bf ? vround(pm) : pm, z5);
    Vec out = vload_vec(a.b6, q);
    const Vec u5 = vsilu(z5 FE_ACT(a));
    gemm64(a.wpack + (size_t)I_W6 * IMG, bf ? vround(u5) : u5, out);
    if (a.flags & FASTEGNN_F_RESIDUAL) vadd(out, hv);
    if (valid) vstore_row(a.HvT_out + (size_t)m * H, q, out);
  }
}
int graph_post_forward(const fastegnn_layer_t *L, hipStream_t st) {
  FE_REQUIRE(L->xsum && L->Z && L->HvT && L->poolV && L->poolX && L->Z_out && L->HvT_out && L->wpack,
             "graph_post_forward: null buffer");
  GraphPostArgs a{L->xsum, L->Z, L->HvT, L->poolV, L->poolX, L->wpack, L->params[FASTEGNN_P_NODEV0_B],
                  L->params[FASTEGNN_P_NODEV2_B], L->Z_out, L->HvT_out, L->B, L->C, L->flags, L->act_param};
  int grid = cdiv(cdiv((long)L->B * L->C, 16), 4);
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  { ProfScope _ps_graph_post_fwd_kernel(K_GRAPH_POST_FWD, st); hipLaunchKernelGGL(graph_post_fwd_kernel, dim3(grid), dim3(256), 0, st, a); }
  return check_launch("graph_post_fwd_kernel");
}

}  // namespace fe

#ifdef FE_STAMP_VF
extern "C" int fastegnn_debug_read_stamps_vf(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::g_vf_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fe::g_vf_stamps), z, sizeof(z));
  }
  return 0;
}
#endif
#ifdef FE_STAMP
extern "C" int fastegnn_debug_read_stamps(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fe::g_stamps), z, sizeof(z));
  }
  return 0;
}
#endif
