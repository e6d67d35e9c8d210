// B4 (virtual stage backward) for 1 <= C <= 64, and since round 4 the FastRF / EGNN wirings too (the round-2 single-kernel
// form and its wgrad bundle are gone): the adjoint of edge_mode_virtual / coord_model_vel (virtual part) /
// coord_model_virtual / node_model (models/FastEGNN.py:111-119,136-166) as three kernels.  Math: oracle/factored.py (virt_bwd).
//
//   virt_bwd_node_kernel  node_mlp adjoint of every node: g_np, g_h (partial), g_aggm, the node-level weight-gradient
//                         operands, the per-node scalars of the coordinate update.
//   virt_bwd_gv_kernel    Gv[c][n] = g_poolV[b,c] + W3c[c]^T g_np[n]: the part of d/dv that does not depend on the
//                         recomputed forward, as one dense product (a workgroup keeps four W3c images resident and
//                         streams its share of the nodes).  Taking it out of the channel loop removes that loop's per-channel
//                         W3c stage (55 KB of LDS, one workgroup barrier per channel, an LDS-DMA pipeline with counted
//                         waits) -- the waves of the main kernel no longer walk the channels in lock step.
//   virt_bwd_pc_kernel    per (16-node tile, channel): forward recompute interleaved with its adjoint, six producer
//                         waves; the three 64x64 weight gradients over the (node, channel) rows -- coord_mlp_r_virtual.0,
//                         coord_mlp_v_virtual.0, edge_mlp_virtual.2 -- are contracted INSIDE the workgroup by two consumer
//                         waves fed through LDS rings (as in edge_bwd_pc_kernel); only `v` still goes to HBM, for the
//                         per-channel node_mlp.0 blocks.  Round 2 stored five [C][N][64] operand arrays here (2.05 GB
//                         per cfg4 launch) and read them back in a bundled weight-gradient kernel.
#include <cstdlib>
#include "stages.h"

namespace fe {

#ifdef FE_SAFE_WAITS   // ring flags as workgroup-scope acquire loads / release stores instead of relaxed accesses between fences
__device__ __forceinline__ int vb_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void vb_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
__device__ __forceinline__ int vb_ld(const int *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void vb_st(int *p, int v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
#endif

// in-kernel phase stamps of virt_bwd_pc_kernel (diagnostic builds only: -DFE_STAMP; tools/gpu_stamp_vb2.py)
#ifdef FE_STAMP
__device__ unsigned long long g_vb2_stamps[32];
#define VB2_T0() unsigned _vp = (unsigned)__builtin_amdgcn_s_memtime(); unsigned _va[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define VB2_T(i) { __builtin_amdgcn_sched_barrier(0); const unsigned _t = (unsigned)__builtin_amdgcn_s_memtime(); \
                   _va[i] += _t - _vp; _vp = _t; __builtin_amdgcn_sched_barrier(0); }
#define VB2_TEND(base) if (lane_id() == 0) { for (int _k = 0; _k < 12; ++_k) atomicAdd(&g_vb2_stamps[(base) + _k], (unsigned long long)_va[_k]); }
#elif defined(FE_ISA_MARK)   // assembly-only builds of tools/isa_budget_bwd.py: the phase boundaries as comments between scheduling fences
#define VB2_T0()
#define VB2_T(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("; FE_MARK " #i); __builtin_amdgcn_sched_barrier(0); }
#define VB2_TEND(base)
#else
#define VB2_T0()
#define VB2_T(i)
#define VB2_TEND(base)
#endif

__device__ __forceinline__ Vec vb_dsilu_mul(const Vec &g, const Vec &z FE_ACT_P) {
  return vmap2(g, z, [=](float a, float b) { return a * dsilu_f(b FE_ACT_A); });
}
__device__ __forceinline__ Vec vb_mask(const Vec &v, bool keep) { return keep ? v : vzero(); }

// =====================================================================================
// B4a node level: adjoint of node_mlp (node_model, :153-166) + per-node scalars of coord_model_vel (:136-142)
// =====================================================================================
struct VirtNodeArgs {
  const float *g_h_out, *npre, *g_x_out, *vel, *aggx, *wpack;
  float *wg_t3, *wg_gnp, *g_h, *g_aggm, *g_aggx, *g_svel, *g_sgrav;
  int N, flags, C;
  float g[3];
  float act_param = 0.f;
};
constexpr int VB_NODE_WAVES = 8;
template <bool BF>
__global__ __launch_bounds__(64 * VB_NODE_WAVES) void virt_bwd_node_kernel(VirtNodeArgs a) {
  constexpr int SM = BF ? GM_BF16 : GM_X3;   // split images: bf16x3 products (or one bf16 product of the rounded operand)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned *img = reinterpret_cast<unsigned *>(lds);
  const bool rf = a.flags & FASTEGNN_F_RF;   // FastRF.py:186: no node_model, the node features pass through
  if (!rf) load_images_x3(img, wpack_x3(a.wpack, a.C, I_W3AT), 3);   // W3AT, W3BT, W4T (consecutive ids)
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4;
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ntiles = (a.N + 15) >> 4;
  const bool clamp_aggx = a.flags & FASTEGNN_F_EGNN;
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int n = tile * 16 + j;
    const bool valid = n < a.N;
    const int nc = valid ? n : a.N - 1;
    const Vec g_out = vb_mask(vload_row(a.g_h_out + (size_t)nc * H, q), valid);
    if (rf) {
      if (valid) {
        vstore_row(a.g_h + (size_t)n * H, q, g_out);
        vstore_row(a.g_aggm + (size_t)n * H, q, vzero());   // the segment-mean message feeds nothing
      }
    } else {
      const Vec npre = vload_row(a.npre + (size_t)nc * H, q);
      Vec g_t3 = vzero();
      gemm_op<SM>(img, 2, make_operand<SM>(g_out), g_t3);
      const Vec g_np = vb_dsilu_mul(g_t3, npre FE_ACT(a));
      if (valid) {
        vstore_row(a.wg_t3 + (size_t)n * H, q, vsilu(npre FE_ACT(a)));
        vstore_row(a.wg_gnp + (size_t)n * H, q, g_np);
      }
      const typename OperandOf<SM>::type gop = make_operand<SM>(g_np);   // shared by the two products
      Vec g_h = (a.flags & FASTEGNN_F_RESIDUAL) ? g_out : vzero();
      gemm_op<SM>(img, 0, gop, g_h);
      if (valid) vstore_row(a.g_h + (size_t)n * H, q, g_h);
      Vec g_am = vzero();
      gemm_op<SM>(img, 1, gop, g_am);
      if (valid) vstore_row(a.g_aggm + (size_t)n * H, q, g_am);
    }
    if (valid && q == 0) {
      float sv = 0.f, sg = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float gxn = a.g_x_out[(size_t)n * 3 + k];
        // clamp(tot_f, -100, 100) of the EGNN baseline passes the gradient only inside the interval
        const bool pass = !clamp_aggx || fabsf(a.aggx[(size_t)n * 3 + k]) <= 100.f;
        a.g_aggx[(size_t)n * 3 + k] = pass ? gxn : 0.f;
        sv += gxn * a.vel[(size_t)n * 3 + k];
        sg += gxn * a.g[k];
      }
      a.g_svel[n] = sv;
      if (a.flags & FASTEGNN_F_GRAVITY) a.g_sgrav[n] = sg;
    }
  }
}

// =====================================================================================
// B4b  Gv[c][n][:] = g_poolV[batch[n], c][:] + W3c[c]^T g_np[n][:]
// (d loss / d v[n,c,:] through node_mlp.0's flat(v) block (:157-158) and node_model_virtual's pool (:170))
// =====================================================================================
struct VirtGvArgs {
  const float *g_np, *g_poolV, *wpack;
  const int32_t *batch;
  float *Gv;
  size_t cstride;   // floats per channel block of Gv
  int N, C, ngroups, nranges;
};
constexpr int VB_GV_CH = 4;      // W3c images resident per workgroup
constexpr int VB_GV_WAVES = 8;
template <bool BF>
__global__ __launch_bounds__(64 * VB_GV_WAVES) void virt_bwd_gv_kernel(VirtGvArgs a) {
  constexpr int SM = BF ? GM_BF16 : GM_X3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char *img = reinterpret_cast<char *>(lds);
  const int grp = (int)blockIdx.x % a.ngroups, range = (int)blockIdx.x / a.ngroups;
  const int c0 = grp * VB_GV_CH, ncv = min(VB_GV_CH, a.C - c0);
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(wpack_rm(a.wpack, a.C, RM_FIXED + c0));   // consecutive slots
    u32x4 *dst = reinterpret_cast<u32x4 *>(img);
    for (int i = threadIdx.x; i < ncv * (RM_BYTES / 16); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4, wv = wave_id();
  const int ntiles = (a.N + 15) >> 4;
  const int t_lo = (int)((long)range * ntiles / a.nranges), t_hi = (int)((long)(range + 1) * ntiles / a.nranges);
  for (int tile = t_lo + wv; tile < t_hi; tile += VB_GV_WAVES) {
    const int n = tile * 16 + j;
    const bool valid = n < a.N;
    const int nc = valid ? n : a.N - 1;
    const int b = a.batch[nc];
    Vec g = vb_mask(vload_row(a.g_np + (size_t)nc * H, q), valid);
    if constexpr (BF) g = vround(g);
    const typename OperandOf<SM>::type op = make_operand<SM>(g);
    const float *pv = a.g_poolV + ((size_t)b * a.C + c0) * H;
#pragma unroll 1
    for (int k = 0; k < ncv; ++k) {
      Vec acc = vload_row(pv + (size_t)k * H, q);
      gemm_rm<SM, true>(img + k * RM_BYTES, op, acc);
#ifdef VB_GV_NT   // measured lever, rejected: non-temporal stores for the Gv rows (0.49 -> 1.39 ms per step)
      if (valid) {
        float *row = a.Gv + (size_t)(c0 + k) * a.cstride + (size_t)n * H;
#pragma unroll
        for (int t = 0; t < 4; ++t) __builtin_nontemporal_store(acc.t[t], reinterpret_cast<f32x4 *>(row + 16 * t + 4 * q));
      }
#else
      if (valid) vstore_row(a.Gv + (size_t)(c0 + k) * a.cstride + (size_t)n * H, q, acc);
#endif
    }
  }
}

// =====================================================================================
// B4c main kernel
// =====================================================================================
// Work units of a workgroup (its producers take them from an LDS ticket counter): whole tiles first, then the last
// VB_FINE_TILES tiles of the workgroup's run in units of VB_GF channels, so that the producers finish within a few channels
// of each other (with whole tiles only, the last tile of the slowest producer was 10 % of the kernel: phase stamps).  The
// channel groups 1.. of a fine tile write their share of g_A / g_x to a small `part` tile; virt_bwd_combine_kernel adds them.
#ifndef FE_VB_FLUSH
#define FE_VB_FLUSH 48
#endif
constexpr int VB_FLUSH = FE_VB_FLUSH;   // tickets (16-row operand sets) a consumer accumulates in registers between two slab updates
constexpr int VB_FINE_TILES = 3;
constexpr int VB_GF = 2;
constexpr int VB_WAVES = 8;
// Waves w and w + 4 of a workgroup share a SIMD.  Consumers: waves 3 and 7 (a SIMD of their own) and wave 6 (beside
// producer wave 2); producers: waves 0, 1, 2, 4, 5.
constexpr int VB_CONS_X = 3, VB_CONS_XX = 7, VB_CONS_V2 = 6;
// Round 5, f16x2 build: the consumers contract on f16x2 products with a sticky scale and 32x32x16 MFMAs (common.h, WgAcc32) -- a quarter
// of the matrix-pipe time and half the operand-split work of the bf16x3 form -- so TWO waves carry the three contractions: wave 3
// takes (g_ux, v) and (g_uX, v) (one split of v serves both), wave 7 takes (g_vp, t); wave 6 becomes the sixth producer.
// Measured on one box each (profiles/r05_lever_cons32.txt): bf16x3 consumers 3.209 ms per step; f16x2 consumers, 5 + 3: 3.131; the same
// with the lane-local scale check (WgScale::update_lazy) 3.056; 6 + 2 with it 2.903 (without it 3.151: wave 3 paced the ring).
// -DFE_VB_CONS32=0 restores the bf16x3 consumers (and with them the 5 + 3 split); -DFE_VB_6P=0 keeps 5 + 3 with the new consumers.
#ifndef FE_VB_CONS32
#define FE_VB_CONS32 1
#endif
#ifndef FE_VB_6P
#define FE_VB_6P 1
#endif
constexpr int VB_RS = 68;                     // row stride of a ring tile
constexpr int VB_TILE = 16 * VB_RS;           // floats per 16 x 64 tile
constexpr int VB_SLOT_A = 3 * VB_TILE;        // g_ux | g_uX | v   (read by consumers X and XX)
constexpr int VB_SLOT_B = 2 * VB_TILE;        // g_vp | t          (read by consumer V2)
constexpr int VB_MAXRING = 6;
// control words (LDS ints): unit ticket, ring heads A / B, per ring slot: filled, drained (ring A: one per consumer)
enum { VBC_UNIT = 0, VBC_HEAD = 1, VBC_FILLED = 4, VBC_DRAINED = 4 + 2 * VB_MAXRING, VBC_CTRL = 4 + 5 * VB_MAXRING };
// rank-1 gradient accumulators of the workgroup in LDS: [w_xv2 | w_xx2 | w_vr | att_w | att_b], in up to VB_MAXBANK banks
// (producer p adds to bank p % nbank: shorter fp32 chains; as many banks as the LDS budget leaves)
constexpr int VB_RACC = 5 * H;
constexpr int VB_MAXBANK = 5;

struct VirtBwd2Args {
  VirtArgs f;
  const float *g_x_out, *g_poolX, *Gv;
  float *g_x, *g_A, *g_Bc, *g_Zp;
  float *gA_part, *gx_part;   // channel groups 1.. of the fine tiles: [grid * VB_FINE_TILES][NGF-1][16][64] and ...[16][4]
  float *wg_v;                // [C][N + pad][64]
  float *cons_scratch;        // [grid][3][64*64] running sums of the three consumers (accumulator order)
  size_t cstride;
  float *d_wxv2, *d_wxx2, *d_wvr, *d_attw, *d_attb;   // rank-1 gradients (d_wvr strided by ld_v0)
  int ld_v0;
  float *slab, *slab_b;
  int slab_x, slab_X, slab_v2;   // first partial slab of coord_mlp_r_virtual.0 / coord_mlp_v_virtual.0 / edge_mlp_virtual.2
  int NGF, ringA, ringB, nbank;   // NGF: channel groups of a fine tile
};

inline size_t vb_lds_floats(int C, int ringA, int ringB, int nbank, int img_bytes = RM_BYTES) {
  return (size_t)3 * (img_bytes / 4) + VV_COUNT * H + (size_t)nbank * VB_RACC + (size_t)C * H + ((3 * C + 3) & ~3) + (size_t)ringA * VB_SLOT_A +
         (size_t)ringB * VB_SLOT_B + VBC_CTRL;
}

__device__ __forceinline__ void vb_tile_store(float *tile, int j, int q, const Vec &v) {
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(tile + j * VB_RS + 16 * t + 4 * q) = v.t[t];
}
// row[o] += sum over the 16 items of the tile of u[o][item]: transposing DPP butterfly, then ONE 64-lane atomic
__device__ __forceinline__ void vb_accum_items(float *row, const Vec &u, int j, int q) {
#ifdef VB_DIAG_NOACC   // diagnostic: what do the DPP sums + LDS atomics of the rank-1 gradients and pools cost?
  if (u.t[0][0] == 12345.678f) row[0] = 1.f;
  return;
#endif
  tile_sum_add(row, u, j, q);
}

template <bool BF, bool ATT>
__global__ __launch_bounds__(64 * VB_WAVES) void virt_bwd_pc_kernel(VirtBwd2Args A) {
  constexpr int SM = BF ? GM_BF16 : GM_VIRT_BWD;
  typedef typename OperandOf<SM>::type SOp;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const VirtArgs &a = A.f;
  const int C = a.C;
  constexpr int RMS = rm_lds_bytes<SM>();               // LDS bytes per image: an f16x2 image has two parts (36.9 instead of 55.3 KB)
  char *rmimg = reinterpret_cast<char *>(lds);          // V2 | WXV0 | WXX0, row-major split images (product and transpose)
  float *vec = lds + 3 * (RMS / 4);
  float *racc0 = vec + VV_COUNT * H;                    // [nbank][5][64]
  float *gBc_l = racc0 + A.nbank * VB_RACC;             // [C][64]
  float *gZ_l = gBc_l + C * H;                          // [3][C]
  float *ringA = gZ_l + ((3 * C + 3) & ~3);
  float *ringB = ringA + A.ringA * VB_SLOT_A;
  int *ctrl = reinterpret_cast<int *>(ringB + A.ringB * VB_SLOT_B);
  {
    // slots 0..2 (f16x2: 7..9) are consecutive in wpack, RM_BYTES apart; only the parts this form reads are copied
    const char *src = wpack_rm(a.wpack, C, SM == GM_F16 ? RM_F16 : 0);
    for (int i = threadIdx.x; i < 3 * (RMS / 16); i += blockDim.x) {
      const int im = i / (RMS / 16), k = i % (RMS / 16);
      reinterpret_cast<u32x4 *>(rmimg + im * RMS)[k] = reinterpret_cast<const u32x4 *>(src + (size_t)im * RM_BYTES)[k];
    }
  }
  virt_load_vecs(vec, a);
  for (int i = threadIdx.x; i < A.nbank * VB_RACC + C * H + 3 * C; i += blockDim.x) racc0[i] = 0.f;
  if (threadIdx.x < VBC_CTRL) ctrl[threadIdx.x] = 0;
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4, wv = wave_id();
  constexpr bool CONS32 = SM == GM_F16 && FE_VB_CONS32 != 0;   // f16x2 / 32x32x16 consumers (the bf16 mode and the x3 build keep theirs)
  constexpr bool SIXP = CONS32 && FE_VB_6P != 0;               // 6 producers + 2 consumers
  const bool consumer = wv == VB_CONS_X || wv == VB_CONS_XX || (!SIXP && wv == VB_CONS_V2);
  const int ntiles = (a.N + 15) >> 4;
  const int t_lo = (int)((long)blockIdx.x * ntiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int n_fine = min(t_hi - t_lo, VB_FINE_TILES), n_coarse = t_hi - t_lo - n_fine;
  const int n_units = n_coarse + n_fine * A.NGF;
#ifdef VB_DIAG_NOPUB   // diagnostic: producers skip the ring hand-offs, consumers have nothing to do
  const int total = 0;
#else
  const int total = (t_hi - t_lo) * C;            // (tile, channel) operand sets = tickets per ring
#endif
  const int cur = a.batch[t_lo * 16];             // graph whose pools this workgroup accumulates in LDS
  const bool tanh_on = a.flags & FASTEGNN_F_TANH;

  if (CONS32 && consumer) {
    if constexpr (CONS32) {
    // ---------------------------------------------------------------------------------------------------------
    // consumers, f16x2 form (common.h: WgAcc32 / WgScale): one ticket (16 rows) per step, no pairing.
    //   SIXP : wave 3 = roles 0 + 1 (X and XX, both from ring A, v split once), wave 7 = role 2 (V2, ring B)
    //   else : wave 3 = role 0, wave 7 = role 1, wave 6 = role 2 (one accumulator each, as the bf16x3 consumers)
    // ---------------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    const bool ringA_wave = SIXP ? wv == VB_CONS_X : wv != VB_CONS_V2;
    const int role0 = SIXP ? (wv == VB_CONS_X ? 0 : 2) : (wv == VB_CONS_X ? 0 : (wv == VB_CONS_XX ? 1 : 2));
    const bool two_acc = SIXP && wv == VB_CONS_X;
    const int RING = ringA_wave ? A.ringA : A.ringB;
    const int slot_f = ringA_wave ? VB_SLOT_A : VB_SLOT_B;
    const float *ring = ringA_wave ? ringA : ringB;
    const int g_off0 = role0 == 1 ? VB_TILE : 0, g_off1 = VB_TILE, t_off = ringA_wave ? 2 * VB_TILE : VB_TILE;
    const int *filled = ctrl + VBC_FILLED + (ringA_wave ? 0 : VB_MAXRING);
    int *drained0 = ctrl + VBC_DRAINED + role0 * VB_MAXRING;
    int *drained1 = ctrl + VBC_DRAINED + 1 * VB_MAXRING;   // second flag of a ring-A slot (set by this wave when it holds both roles)
    WgAcc32 acc0, acc1;
    wg32_zero(acc0);
    wg32_zero(acc1);
    WgScale sG0{0}, sG1{0}, sT{0};
    double bs0[2] = {0., 0.}, bs1[2] = {0., 0.};          // bias column sums of this lane's rows: feature 32 b + i, double across tickets
    // running sums in TRUE units, accumulator order, in this wave's scratch tiles (L2 resident): [role][block (bo, bk)][16 regs][64 lanes]
    auto scp = [&](int role, int blk, int e4) {
      char *b = reinterpret_cast<char *>(A.cons_scratch + ((size_t)blockIdx.x * 3 + role) * IMG) + (size_t)((blk * 4 + e4) * 64 * 16);
      asm volatile("" : "+s"(b));
      return reinterpret_cast<f32x4 *>(b + (unsigned)l * 16u);
    };
    bool flushed = false;
    auto flush_one = [&](WgAcc32 &acc, const WgScale &sg, int role, bool last, float *slab_dst) {
      const float ig = sg.inv(), it = sT.inv();
#pragma unroll
      for (int bo = 0; bo < 2; ++bo)
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (acc.c[bo][bk][4 * e4 + r] * ig) * it;
            f32x4 *d = scp(role, bo * 2 + bk, e4);
            if (flushed) v += *d;
            if (!last) {
              *d = v;
            } else {
              // [o][k] row-major slab: o = 32 bo + 8 e4 + 4 half + r, k = 32 bk + (lane & 31)
#pragma unroll
              for (int r = 0; r < 4; ++r) slab_dst[(32 * bo + 8 * e4 + 4 * (l >> 5) + r) * H + 32 * bk + (l & 31)] = v[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.c[bo][bk][4 * e4 + r] = 0.f;
          }
    };
    int since = 0;
    for (int done = 0; done < total; ++done) {
      const int s0 = done % RING, r0w = done / RING;
      while (vb_ld(&filled[s0]) != r0w + 1) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // the slot reads stay behind the flag read
      const float *g0 = ring + s0 * slot_f;
      float xt[2][8], xg0[2][8], xg1[2][8];
      wg32_read<VB_RS>(g0 + t_off, xt);
      wg32_read<VB_RS>(g0 + g_off0, xg0);
      if (two_acc) wg32_read<VB_RS>(g0 + g_off1, xg1);
      // every read of the slot has returned (lgkmcnt(0)) and none of them may sink below the hand-back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (l == 0) {
        vb_st(&drained0[s0], r0w + 1);
        if (two_acc) vb_st(&drained1[s0], r0w + 1);
      }
      // sticky scales: a larger tile lowers the stream's scale and rescales what the accumulators hold (rare, wave-uniform)
      {
        const float fT = sT.update_lazy(xt);
        const float f0 = sG0.update_lazy(xg0) * fT;
        if (f0 != 1.f) wg32_scale_acc(acc0, f0);
        if (two_acc) {
          const float f1 = sG1.update_lazy(xg1) * fT;
          if (f1 != 1.f) wg32_scale_acc(acc1, f1);
        }
      }
      const WgOp32 T = wg32_split(xt, sT.scale());
      {
        const WgOp32 G = wg32_split(xg0, sG0.scale());
#pragma unroll
        for (int b = 0; b < 2; ++b)
          bs0[b] += (double)(((xg0[b][0] + xg0[b][1]) + (xg0[b][2] + xg0[b][3])) + ((xg0[b][4] + xg0[b][5]) + (xg0[b][6] + xg0[b][7])));
        wg32_mma(acc0, G, T);
      }
      if (two_acc) {
        const WgOp32 G = wg32_split(xg1, sG1.scale());
#pragma unroll
        for (int b = 0; b < 2; ++b)
          bs1[b] += (double)(((xg1[b][0] + xg1[b][1]) + (xg1[b][2] + xg1[b][3])) + ((xg1[b][4] + xg1[b][5]) + (xg1[b][6] + xg1[b][7])));
        wg32_mma(acc1, G, T);
      }
      if (++since >= VB_FLUSH && done + 1 < total) {   // the accumulators leave the registers every VB_FLUSH tickets (cancelling sums)
        flush_one(acc0, sG0, role0, false, nullptr);
        if (two_acc) flush_one(acc1, sG1, 1, false, nullptr);
        flushed = true;
        since = 0;
      }
    }
    // one partial slab per workgroup and weight ([o][k] row-major) + its bias column sums
    auto finish = [&](WgAcc32 &acc, const WgScale &sg, int role, double (&bs)[2]) {
      const size_t sl = (size_t)(role == 0 ? A.slab_x : (role == 1 ? A.slab_X : A.slab_v2)) + blockIdx.x;
      flush_one(acc, sg, role, true, A.slab + sl * IMG);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        double s0 = bs[b];
        s0 += __shfl_xor(s0, 32);   // the two row halves of the feature
        if (l < 32) A.slab_b[sl * H + 32 * b + l] = (float)s0;
      }
    };
    finish(acc0, sG0, role0, bs0);
    if (two_acc) finish(acc1, sG1, 1, bs1);
    }
#ifdef VB_NO_CONS
  } else if (false) {
#else
  } else if (consumer) {
#endif
    // ---------------------------------------------------------------------------------------------------------
    // consumers (as in edge_bwd_pc_kernel): one 64x64 accumulator each, two tickets of the ring per step (K = 32 rows),
    // operands split into bf16 parts here, six bf16x3 products per 16x16 tile (one in bf16 mode); the slots are handed
    // back as soon as their values are in registers.  X: (g_ux, v), XX: (g_uX, v) -- both read ring A, each has its own
    // drained flag per slot --, V2: (g_vp, t) from ring B.
    // ---------------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    const int role = wv == VB_CONS_X ? 0 : (wv == VB_CONS_XX ? 1 : 2);
    const int RING = role < 2 ? A.ringA : A.ringB;
    const int slot_f = role < 2 ? VB_SLOT_A : VB_SLOT_B;
    const float *ring = role < 2 ? ringA : ringB;
    const int g_off = role == 1 ? VB_TILE : 0, t_off = role < 2 ? 2 * VB_TILE : VB_TILE;
    const int *filled = ctrl + VBC_FILLED + (role < 2 ? 0 : VB_MAXRING);
    int *drained = ctrl + VBC_DRAINED + role * VB_MAXRING;
    f32x4 acc[4][4];
    // column sums of G (bias gradient), two levels: fp32 over the 768 rows between two flushes, DOUBLE across the flushes (the
    // layer-0 bias sums of the coordinate heads cancel to ~1e-3 of their terms over a workgroup's 6 144 / 12 288 rows)
    double bs[4] = {0., 0., 0., 0.};      // (eight rows at a time in fp32, then double: 1.46e-5 against a tolerance of 1.21e-5 on
    double bs_tot[4] = {0., 0., 0., 0.};   //  edge_mlp_virtual.2.bias at C = 48 with fp32 chains of 768 rows)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tk = 0; tk < 4; ++tk) acc[ti][tk] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The accumulator leaves the registers every VB_FLUSH tickets (768 rows): one fp32 chain over all of a workgroup's rows
    // (6 144 at C = 16, 12 288 at C = 32) measured 2-4x the rounding noise of round 2's 780-row chains on the cancelling
    // layer-0 sums (test_cfg5_shape_c32_vs_oracle).  The slab is this wave's own: plain read-modify-write, L2 resident.
    // (The running sum lives in a scratch tile of this wave in ACCUMULATOR order -- sixteen 16-byte read-modify-writes per
    // lane off one base address, L2 resident; the [o][k] slab is written once, at the end.)
    // (wave-uniform base + constant on the scalar side, ONE 32-bit lane offset: per-tile 64-bit lane pointers get hoisted
    // out of the ticket loop and spilled)
    char *scb = reinterpret_cast<char *>(A.cons_scratch + ((size_t)blockIdx.x * 3 + role) * IMG);
    const unsigned sco = (unsigned)l * 16u;
    auto scp = [&](int ti, int tk) {
      char *b = scb + (size_t)((ti * 4 + tk) * 64 * 16);
      asm volatile("" : "+s"(b));
      return reinterpret_cast<f32x4 *>(b + sco);
    };
    bool flushed = false;
    auto flush = [&]() {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tk = 0; tk < 4; ++tk) {
          f32x4 *d = scp(ti, tk);
          if (flushed) acc[ti][tk] += *d;
          *d = acc[ti][tk];
          acc[ti][tk] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        bs_tot[ti] += bs[ti];
        bs[ti] = 0.;
      }
      flushed = true;
    };
    int done = 0, since = 0;
    VB2_T0()
    while (done < total) {
      VB2_T(2)
      const int s0 = done % RING, r0w = done / RING;
      const bool two = done + 1 < total;
      const int s1 = (done + 1) % RING, r1w = (done + 1) / RING;
      while (vb_ld(&filled[s0]) != r0w + 1) __builtin_amdgcn_s_sleep(1);
      if (two)
        while (vb_ld(&filled[s1]) != r1w + 1) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // the slot reads stay behind the flag reads
      VB2_T(0)   // consumer: waiting for filled slots
      const float *g0 = ring + s0 * slot_f, *g1 = ring + s1 * slot_f;
      // lane (q,i) takes feature 16t + i of the rows 4q + e of the first (e < 4) and of the second slot (e >= 4), the same
      // map for both operands; with the 68-float row stride the reads are conflict-free
      float xb[4][8], xa[4][8];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xb[t][e] = g0[t_off + (4 * q + e) * VB_RS + 16 * t + j];
          xb[t][4 + e] = two ? g1[t_off + (4 * q + e) * VB_RS + 16 * t + j] : 0.f;
          xa[t][e] = g0[g_off + (4 * q + e) * VB_RS + 16 * t + j];
          xa[t][4 + e] = two ? g1[g_off + (4 * q + e) * VB_RS + 16 * t + j] : 0.f;
        }
      // every read of the slots has returned (lgkmcnt(0)) and, for the compiler, none of them may sink below the hand-back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (l == 0) {
        vb_st(&drained[s0], r0w + 1);
        if (two) vb_st(&drained[s1], r1w + 1);
      }
      Split8 Bop[4];   // bf16 mode: only .h is used (RNE-rounded operand, one product)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (BF) Bop[t].h = round8(xb[t]);
        else Bop[t] = split8(xb[t]);
      }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const float (&x)[8] = xa[ti];
        bs[ti] += (double)(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7])));
        if constexpr (BF) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, round8(x));
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[ti][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, Bop[t].h), acc[ti][t], 0, 0, 0);
        } else {
          const Split8 Aop = split8(x);
          const bf16x8 ah = __builtin_bit_cast(bf16x8, Aop.h), am = __builtin_bit_cast(bf16x8, Aop.m),
                       al = __builtin_bit_cast(bf16x8, Aop.l);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, Bop[t].h), bm = __builtin_bit_cast(bf16x8, Bop[t].m),
                         bl = __builtin_bit_cast(bf16x8, Bop[t].l);
            f32x4 c = acc[ti][t];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
            acc[ti][t] = c;
          }
        }
      }
      done += 2;
      since += 2;
      if (since >= VB_FLUSH && done < total) {
        flush();
        since = 0;
      }
      VB2_T(1)   // consumer: reads, splits, products
    }
    VB2_TEND(16 + 4 * role)
    // one partial slab per workgroup and weight: [o][k] row-major, o = G feature, k = T feature
    const size_t sl = (size_t)(role == 0 ? A.slab_x : (role == 1 ? A.slab_X : A.slab_v2)) + blockIdx.x;
    float *sa = A.slab + sl * IMG;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tk = 0; tk < 4; ++tk) {
        f32x4 v = acc[ti][tk];
        if (flushed) v += *scp(ti, tk);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          char *b = reinterpret_cast<char *>(sa) + (size_t)(((16 * ti + r) * H + 16 * tk) * 4);
          asm volatile("" : "+s"(b));
          *reinterpret_cast<float *>(b + (unsigned)((4 * q * H + j) * 4)) = v[r];
        }
      }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      double s0 = bs_tot[ti] + bs[ti];   // sum over the four q-lanes of the feature, in double
      s0 += __shfl_xor(s0, 16);
      s0 += __shfl_xor(s0, 32);
      if (q == 0) A.slab_b[sl * H + 16 * ti + j] = (float)s0;
    }
#ifdef VB_NO_PROD
  } else if (false) {
#else
  } else {
#endif
    // ---------------------------------------------------------------------------------------------------------
    // producers: forward recompute of (tile, channel) interleaved with its adjoint.  Register budget: 256 per wave
    // (two waves per SIMD), so the channel-invariant rows are re-read per channel (L2), the tile's g_A accumulates
    // through memory (the rows stay in L2 between the channels of a unit) and the rank-1 gradients leave the wave per
    // channel (DPP sum over the tile, LDS atomics) instead of living in per-lane accumulators.
    // ---------------------------------------------------------------------------------------------------------
    const float invC = 1.0f / (float)C;
    const float attb0 = ATT ? a.attb[0] : 0.f;
    float *racc = racc0 + ((wv > 3 ? wv - 1 : wv) % A.nbank) * VB_RACC;   // this producer's bank (producers: waves 0,1,2,4,5)
#ifdef VB_PRIO
#define VB_PRIO_ON() __builtin_amdgcn_s_setprio(VB_PRIO)
#define VB_PRIO_OFF() __builtin_amdgcn_s_setprio(0)
#else
#define VB_PRIO_ON()
#define VB_PRIO_OFF()
#endif
    // (P16 = false: this kernel sits at 256 registers with spills; the pipelined f16x2 product costs it more than it hides, common.h)
    auto mm = [&](int which, const SOp &op, Vec &acc) { VB_PRIO_ON(); gemm_rm<SM, false, true, false>(rmimg + which * RMS, op, acc); VB_PRIO_OFF(); };
    auto mmT = [&](int which, const Vec &g, Vec &acc) {
      const auto op = make_grad_operand<SM>(g);   // (the f16x2 form scales a gradient per item)
      VB_PRIO_ON();
      gemm_rm_g<SM, true, true, false>(rmimg + which * RMS, op, acc);
      VB_PRIO_OFF();
    };
    VB2_T0()
    for (;;) {
      int u = 0;
      if (l == 0) u = atomicAdd(&ctrl[VBC_UNIT], 1);
      u = __builtin_amdgcn_readfirstlane(u);
      if (u >= n_units) break;
      int tile, grp, c_lo, c_hi, fslot = 0;
      if (u < n_coarse) {
        tile = t_lo + u; grp = 0; c_lo = 0; c_hi = C;
      } else {
        const int uf = u - n_coarse;
        fslot = uf / A.NGF; grp = uf % A.NGF;
        tile = t_lo + n_coarse + fslot;
        c_lo = grp * VB_GF; c_hi = min(C, c_lo + VB_GF);
      }
      const int n0 = tile * 16, nend = min(a.N, n0 + 16);
      const int b0 = a.batch[n0], b1 = a.batch[nend - 1];
      // pools of this unit: 0 = the workgroup's LDS accumulators (the tile lies in graph `cur`), 1 = one global atomic per
      // (channel, feature) after the sum over the tile (the tile lies in one other graph), 2 = one atomic per row
      const int pmode = b0 != b1 ? 2 : (b0 == cur ? 0 : 1);
      const int n = n0 + j;
      const bool valid = n < nend;
      const int nc = valid ? n : nend - 1;
      const int b = a.batch[nc];
      const unsigned offB = (unsigned)b * C * H + 4u * q;                // [B,C,64] arrays (+ c*H)
      const unsigned offN = (unsigned)nc * H + 4u * q;                   // [N,64] arrays
      float gxn[3], xi[3], gx[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gxn[k] = valid ? A.g_x_out[(size_t)nc * 3 + k] : 0.f;
        xi[k] = a.x[(size_t)nc * 3 + k];
        gx[k] = grp == 0 ? gxn[k] : 0.f;     // the direct path x' = x + ... is counted once per tile
      }
      const unsigned offZ = (unsigned)b * 3u * C;                        // [B,3,C] arrays (+ k*C + c)
      // where this unit's share of g_A / g_x goes: the arrays themselves, or a part tile (channel groups 1.. of a fine tile)
      const size_t part = ((size_t)blockIdx.x * VB_FINE_TILES + fslot) * (A.NGF - 1) + (grp - 1);
      float *dA = grp == 0 ? A.g_A : A.gA_part + part * 16 * H;
      const unsigned offA = grp == 0 ? offN : (unsigned)j * H + 4u * q;
      // the tile's g_A stays in 16 registers across the channels of the unit and is stored once (round 4; round 3 accumulated it
      // through memory per channel: with the bf16x3 operands there was no room -- now 9 spilled registers, virt_bwd 3.21 -> 3.10 ms
      // per step, 1.30 -> 1.11 GB of HBM traffic per launch; -DVB_GA_MEM restores the old form)
#ifndef VB_GA_MEM
      Vec ga_acc = vzero();
#endif
#pragma unroll 1
      for (int c = c_lo; c < c_hi; ++c) {
        asm volatile("" ::: "memory");
        VB2_T(0)   // unit / channel bookkeeping
        const size_t cb = (size_t)c * A.cstride;
        float vd[3], gpX[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          vd[k] = (a.Z + (k * C + c))[offZ] - xi[k];          // wave-uniform base + 32-bit lane offset (no 64-bit lane pointers)
          gpX[k] = valid ? (A.g_poolX + (k * C + c))[offZ] : 0.f;
        }
        const float vr = sqrt_f(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]);
        Vec vp = vload_vec(vec + VV_C2 * H, q);
        // (requesting these two rows a channel ahead costs 32 registers across the whole body: 57 more spilled registers
        // at 256 per wave for ~1 k of 31 k cycles -- not done)
        Vec d_pre = vload_u(a.A, offN);
        vadd(d_pre, vload_u(a.Bc, offB + (unsigned)c * H));
        vaxpy(d_pre, vr, vload_vec(vec + VV_WVR * H, q));
        VB2_T(1)   // rows arrived, pre formed
        const Vec t = vsilu_keep_d(d_pre FE_ACT(a));        // d_pre <- silu'(pre)
        mm(0, make_operand<SM>(t), vp);
        VB2_T(2)   // silu, split, V2 product
        const Vec v0 = vsilu_keep_d(vp FE_ACT(a));          // vp <- silu'(vp)
        float att = 1.f;
        Vec v = v0;
        if constexpr (ATT) {
          att = sigmoid_f(vdot(v0, vload_vec(vec + VV_ATT * H, q)) + attb0);
          v = vscale(v0, att);
        }
        VB2_T(3)   // silu
        float sx, sX;
        Vec g_ux, g_uX;
        {
          // both head products first: their shared operand (48 registers in the bf16x3 form) is dead before the activations
          Vec uxp = vload_vec(vec + VV_BXV0 * H, q), uXp = vload_vec(vec + VV_BXX0 * H, q);
          {
            const SOp vs = make_operand<SM>(v);
            mm(1, vs, uxp);
            mm(2, vs, uXp);
          }
          {  // coord_mlp_r_virtual head: activation, scalar head, adjoint of the activation
            Vec ux = vsilu_keep_d(uxp FE_ACT(a));             // uxp <- silu'(uxp)
            const float sr = vdot(ux, vload_vec(vec + VV_WXV2 * H, q));
            sx = tanh_on ? tanh_f(sr) : sr;
            float g_sx = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) g_sx -= vd[k] * invC * gxn[k];
            const float g_sr = tanh_on ? g_sx * (1.f - sx * sx) : g_sx;
            vb_accum_items(racc + 0 * H, vscale(ux, g_sr), j, q);
            g_ux = vmul(vscale(vload_vec(vec + VV_WXV2 * H, q), g_sr), uxp);
          }
          VB2_T(4)   // both head products + head x activation / rank-1 sum
          {  // coord_mlp_v_virtual head
            Vec uX = vsilu_keep_d(uXp FE_ACT(a));
            const float sr = vdot(uX, vload_vec(vec + VV_WXX2 * H, q));
            sX = tanh_on ? tanh_f(sr) : sr;
            float g_sX = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) g_sX += vd[k] * gpX[k];
            const float g_sr = tanh_on ? g_sX * (1.f - sX * sX) : g_sX;
            vb_accum_items(racc + 1 * H, vscale(uX, g_sr), j, q);
            g_uX = vmul(vscale(vload_vec(vec + VV_WXX2 * H, q), g_sr), uXp);
          }
        }
        VB2_T(5)   // head X forward + rank-1 sum
        Vec g_v = vload_u(A.Gv + cb, offN);   // (requested ahead of the publish: it arrives under the slot wait)
        // v[c][n] takes the place of Gv[c][n]: ONE [C][N][64] array serves both (the row has just been requested by this
        // very lane; same-address accesses of a lane stay in program order)
        if (valid) vstore_u(A.wg_v + cb, offN, v);
#ifndef VB_DIAG_NOPUB
        {   // (g_ux, v) and (g_uX, v) to consumers X and XX: one slot of ring A, free once both have drained it
          int tk = 0;
          if (l == 0) tk = atomicAdd(&ctrl[VBC_HEAD + 0], 1);
          tk = __builtin_amdgcn_readfirstlane(tk);
          const int sl = tk % A.ringA, round = tk / A.ringA;
          while (vb_ld(&ctrl[VBC_DRAINED + 0 * VB_MAXRING + sl]) != round || vb_ld(&ctrl[VBC_DRAINED + 1 * VB_MAXRING + sl]) != round)
            __builtin_amdgcn_s_sleep(2);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // the tile stores stay behind the flag reads
          float *slot = ringA + sl * VB_SLOT_A;
          vb_tile_store(slot, j, q, g_ux);
          vb_tile_store(slot + VB_TILE, j, q, g_uX);
          vb_tile_store(slot + 2 * VB_TILE, j, q, v);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // lgkmcnt(0): the tiles are in LDS before the flag
          if (l == 0) vb_st(&ctrl[VBC_FILLED + sl], round + 1);
        }
#endif
        VB2_T(6)   // publish to ring A
        g_v = vb_mask(g_v, valid);
        mmT(1, g_ux, g_v);
        mmT(2, g_uX, g_v);
        VB2_T(7)   // Gv row + two transposed head products
        float g_vd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) g_vd[k] = -sx * invC * gxn[k] + sX * gpX[k];
        Vec g_v0 = g_v;
        if constexpr (ATT) {
          const float g_a = vdot(g_v, v0);
          const float g_z = g_a * att * (1.f - att);
          vb_accum_items(racc + 3 * H, vscale(v0, g_z), j, q);
          const float sz = jsum(q == 0 ? g_z : 0.f);
          if (l == 0) atomicAdd(&racc[4 * H], sz);
          g_v0 = vscale(g_v, att);
          vaxpy(g_v0, g_z, vload_vec(vec + VV_ATT * H, q));
        }
        Vec g_t = vzero();
#ifdef VB_GA_MEM
        Vec ga = vzero();
#endif
        {
          const Vec g_vp = vmul(g_v0, vp);
#ifndef VB_DIAG_NOPUB
          {   // (g_vp, t) to consumer V2
            int tk = 0;
            if (l == 0) tk = atomicAdd(&ctrl[VBC_HEAD + 1], 1);
            tk = __builtin_amdgcn_readfirstlane(tk);
            const int sl = tk % A.ringB, round = tk / A.ringB;
            while (vb_ld(&ctrl[VBC_DRAINED + 2 * VB_MAXRING + sl]) != round) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            float *slot = ringB + sl * VB_SLOT_B;
            vb_tile_store(slot, j, q, g_vp);
            vb_tile_store(slot + VB_TILE, j, q, t);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (l == 0) vb_st(&ctrl[VBC_FILLED + VB_MAXRING + sl], round + 1);
          }
#endif
          VB2_T(8)   // g_vp + publish to ring B
          // requested here, consumed after the product: the tile's running g_A
#ifdef VB_GA_MEM
          if (c > c_lo) ga = vload_u(dA, offA);
#endif
          mmT(0, g_vp, g_t);
        }
        VB2_T(9)   // V2^T product
        const Vec g_pre = vmul(g_t, d_pre);
#ifndef VB_GA_MEM
        vadd(ga_acc, g_pre);
        if (c + 1 == c_hi && valid) vstore_u(dA, offA, ga_acc);
#else
        vadd(ga, g_pre);   // g_A of the tile accumulates through memory (the same lane re-reads its own row)
        if (valid) vstore_u(dA, offA, ga);
#endif
        vb_accum_items(racc + 2 * H, vscale(g_pre, vr), j, q);
        const float g_vr = vdot(g_pre, vload_vec(vec + VV_WVR * H, q));
        const float ivr = vr > 0.f ? g_vr * rcp_f(vr) : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          g_vd[k] += ivr * vd[k];
          gx[k] -= g_vd[k];
        }
        // pools over the nodes of the tile: g_Bc[b,c,:] += g_pre, g_Zp[b,:,c] += g_vd (g_pre of a masked lane is zero)
        if (pmode < 2) {
          float pz[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) pz[k] = jsum((valid && q == 0) ? g_vd[k] : 0.f);
          vb_accum_items(pmode == 0 ? gBc_l + c * H : A.g_Bc + ((size_t)b0 * C + c) * H, g_pre, j, q);
          // (lane k < 3 adds component k: per-lane addresses, so the compiler's uniform-address atomic combiner -- a loop over
          // the active lanes per atomic -- stays out)
          if (l < 3) {
            const float pzl = l == 0 ? pz[0] : (l == 1 ? pz[1] : pz[2]);
            if (pmode == 0) atomicAdd(&gZ_l[l * C + c], pzl);
            else atomicAdd(&A.g_Zp[((size_t)b0 * 3 + l) * C + c], pzl);
          }
        } else {
          if (valid && q == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) atomicAdd(&A.g_Zp[((size_t)b * 3 + k) * C + c], g_vd[k]);
          }
          if (valid) {
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                atomicAdd(&A.g_Bc[((size_t)b * C + c) * H + 16 * t2 + 4 * q + r], g_pre.t[t2][r]);
          }
        }
        VB2_T(10)   // g_pre consumers: g_A through memory, w_vr, pools
      }
      if (valid && q == 0) {
        if (grp == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) A.g_x[(size_t)n * 3 + k] = gx[k];
        } else {
          *reinterpret_cast<f32x4 *>(A.gx_part + (part * 16 + j) * 4) = f32x4{gx[0], gx[1], gx[2], 0.f};
        }
      }
    }
    VB2_TEND(0)
  }
  VB2_T0()
  __syncthreads();
  VB2_T(11)   // wait for the rest of the workgroup
  VB2_TEND(consumer ? 12 : 0)
  // per-graph pools of graph `cur` and the rank-1 weight gradients: one atomic set per workgroup
  for (int i = threadIdx.x; i < C * H; i += blockDim.x) atomicAdd(&A.g_Bc[(size_t)cur * C * H + i], gBc_l[i]);
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) atomicAdd(&A.g_Zp[(size_t)cur * 3 * C + i], gZ_l[i]);
  if (threadIdx.x < H) {
    const int o = threadIdx.x;
    float s5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int bk = 0; bk < A.nbank; ++bk) {
      const float *r = racc0 + bk * VB_RACC;
#pragma unroll
      for (int k = 0; k < 4; ++k) s5[k] += r[k * H + o];
      s5[4] += r[4 * H];
    }
    atomicAdd(&A.d_wxv2[o], s5[0]);
    atomicAdd(&A.d_wxx2[o], s5[1]);
    atomicAdd(&A.d_wvr[(size_t)o * A.ld_v0], s5[2]);   // w_vr is column 2H of edge_mlp_virtual.0.weight
    if constexpr (ATT) {
      atomicAdd(&A.d_attw[o], s5[3]);
      if (o == 0) atomicAdd(A.d_attb, s5[4]);
    }
  }
}

// g_A += parts, g_x += parts over the fine tiles of every workgroup of virt_bwd_pc_kernel (same tile ranges), fixed order
__global__ __launch_bounds__(256) void virt_bwd_combine_kernel(float *g_A, float *g_x, const float *gA_part, const float *gx_part,
                                                               int N, int nparts, int grid_pc) {
  const int wg = (int)blockIdx.x / VB_FINE_TILES, fslot = (int)blockIdx.x % VB_FINE_TILES;
  const int ntiles = (N + 15) >> 4;
  const int t_lo = (int)((long)wg * ntiles / grid_pc), t_hi = (int)((long)(wg + 1) * ntiles / grid_pc);
  const int n_fine = min(t_hi - t_lo, VB_FINE_TILES);
  if (fslot >= n_fine) return;
  const int tile = t_hi - n_fine + fslot;
  const size_t p0 = (size_t)blockIdx.x * nparts;
  const int row = threadIdx.x >> 4, c4 = threadIdx.x & 15;   // 16 rows x 16 float4
  const int n = tile * 16 + row;
  if (n >= N) return;
  f32x4 s = reinterpret_cast<const f32x4 *>(g_A + (size_t)n * H)[c4];
  for (int p = 0; p < nparts; ++p) s += reinterpret_cast<const f32x4 *>(gA_part + ((p0 + p) * 16 + row) * H)[c4];
  reinterpret_cast<f32x4 *>(g_A + (size_t)n * H)[c4] = s;
  if (c4 < 3) {
    float sx = g_x[(size_t)n * 3 + c4];
    for (int p = 0; p < nparts; ++p) sx += gx_part[((p0 + p) * 16 + row) * 4 + c4];
    g_x[(size_t)n * 3 + c4] = sx;
  }
}

// =====================================================================================
// B4c', channel-PHASED form (round 5): Gv and v never reach HBM
// =====================================================================================
// The form above keeps two [C][N][64] arrays in HBM per launch: Gv = g_poolV + W3c^T g_np (written by virt_bwd_gv_kernel, read back
// here) and v (written here, read back by wgrad_tn for the per-channel node_mlp.0 blocks) -- 1.6 GB of a launch's 1.1 + 0.5 GB of
// traffic at cfg4, two extra kernels (0.49 + 0.48 ms per step).  Both need ONE 64x64 object per channel -- the image of W3c[c]^T, the
// accumulator of dW3c[c] -- which a tile-major walk over 16 channels cannot keep on chip.  Here a workgroup keeps its tile range and
// walks it CHANNEL-major: units are (channel c, tile) pairs taken from a ticket counter in that order, so at any time the producers
// work on one channel (two around a boundary):
//   * W3c[c]^T sits in an LDS stage of two slots (slot c & 1); the wave that finishes the last unit of channel c - 1 refills that
//     slot with channel c + 1 while the others work on channel c -- no workgroup barrier, one flag per channel;
//   * Gv is formed in registers: g_poolV[b, c] + W3c[c]^T g_np (one more f16x2 product per unit);
//   * the consumer that owns dW3c holds ONE accumulator for the channel in flight (flushed to its scratch tile when a ticket of
//     the other parity arrives, written out as that channel's partial slab when the channel's last ticket has been contracted);
//     g_np rides along in the ring slot of (g_ux, g_uX, v);
//   * g_A / g_x of a tile accumulate through memory from channel to channel (read-modify-write of the tile's rows by whichever
//     producer holds the unit: workgroup-scope release / acquire at unit boundaries; a tile recurs every ntiles-per-workgroup units, far
//     outside the window of units in flight -- the launcher requires >= 12 tiles per workgroup);
//   * pools and the rank-1 gradients as above, the LDS pool rows by channel parity, flushed by the wave that closes the channel.
// Waves: producers 0, 1, 2, 4, 5; wave 3 contracts (g_ux, v) and (g_uX, v), wave 7 (g_vp, t), wave 6 (g_np, v).
// Everything a workgroup needs from its neighbours is gone: no part tiles, no combine kernel, no atomics on g_A.
#ifndef FE_VB_CS
#define FE_VB_CS 1
#endif
#ifndef FE_VBS_5P
#define FE_VBS_5P 1
#endif
constexpr int VBS_SLOT_A = 4 * VB_TILE;        // g_ux | g_uX | v | g_np
constexpr int VBS_W3_WORDS = 4096;             // one f16x2 image of W3c^T (img3 layout, parts h | l)
constexpr int VBS_MIN_TILES = 12;              // tiles per workgroup below which the tile-major form runs instead
constexpr int VBS_MAXC = 64;
constexpr int VBS_MAXTILES = 256;              // tiles per workgroup the per-tile sequence flags cover (1 M nodes on 256 workgroups)
// A long tile range is walked in BLOCKS of VBS_BLOCK tiles (VirtCsArgs::block), channel-major inside a block: every channel pass re-reads the block's A /
// g_np / g_A rows (12 KB per tile), and they must still be in the L2 / Infinity Cache when the next pass comes (cfg5, 244 tiles per
// workgroup, un-blocked: 770 MB of rows re-streamed from HBM 32 times -- 77.7 against 72.2 ms per step of the tile-major form).  A
// PHASE is one (block, channel) pair; phases alternate between the two stage slots / pool rows like channels did.
#ifndef FE_VBS_BLOCK
#define FE_VBS_BLOCK 24
#endif
constexpr int VBS_BLOCK = FE_VBS_BLOCK;
constexpr int VBS_NPH = 64;                    // phase flags live modulo this (at most three phases are in flight)
// control words: unit ticket, ring heads A / B | per ring slot: filled, drained (ring A: two readers), channel of a ring-A slot |
// per channel: image ready, units done
enum { VBSC_UNIT = 0, VBSC_HEAD = 1, VBSC_FILLED = 4, VBSC_DRAINED = 4 + 2 * VB_MAXRING, VBSC_SLOTCH = 4 + 5 * VB_MAXRING,
       VBSC_READY = 4 + 6 * VB_MAXRING, VBSC_DONE = 4 + 6 * VB_MAXRING + VBS_NPH, VBSC_TSEQ = 4 + 6 * VB_MAXRING + 2 * VBS_NPH,
       VBSC_CTRL = 4 + 6 * VB_MAXRING + 2 * VBS_NPH + VBS_MAXTILES };

struct VirtCsArgs {
  VirtArgs f;
  const float *g_x_out, *g_poolX, *g_poolV, *g_np;
  float *g_x, *g_A, *g_Bc, *g_Zp;
  float *cons_scratch;        // [grid][5][64*64]: running sums of X, XX, V2 and of the two parities of W (accumulator order)
  float *w_slab;              // [C][grid][64*64] partial sums of dW3c per workgroup, in ACCUMULATOR order (WgAcc32), summed over the blocks
  float *d_wxv2, *d_wxx2, *d_wvr, *d_attw, *d_attb;
  int ld_v0;
  float *slab, *slab_b;
  int slab_x, slab_X, slab_v2;
  int ringA, ringB, nbank;
  int block;                  // tiles per block of the walk (VBS_BLOCK; FASTEGNN_VIRT_CS_BLOCK overrides it -- tests)
};
inline size_t vbs_lds_floats(int ringA, int ringB, int nbank) {
  return (size_t)3 * (rm_lds_bytes<GM_F16>() / 4) + 2 * VBS_W3_WORDS + VV_COUNT * H + (size_t)nbank * VB_RACC + 2 * H + 8 +
         (size_t)ringA * VBS_SLOT_A + (size_t)ringB * VB_SLOT_B + VBSC_CTRL;
}

// acc += W x on an f16x2 image (img3 layout) with a per-item scaled gradient operand (Split2s): as gemm64_f2_rm_<.., true>
__device__ __forceinline__ void gemm64_f2_scaled(const unsigned *img3, const Split2s &in, Vec &acc) {
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(img3) + lane_id();
  const f16x8 xh0 = __builtin_bit_cast(f16x8, in.s.p[0][0]), xh1 = __builtin_bit_cast(f16x8, in.s.p[0][1]);
  const f16x8 xl0 = __builtin_bit_cast(f16x8, in.s.p[1][0]), xl1 = __builtin_bit_cast(f16x8, in.s.p[1][1]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const f16x8 ah0 = __builtin_bit_cast(f16x8, ip[(t * 2 + 0) * 64]), ah1 = __builtin_bit_cast(f16x8, ip[(t * 2 + 1) * 64]);
    const f16x8 al0 = __builtin_bit_cast(f16x8, ip[512 + (t * 2 + 0) * 64]), al1 = __builtin_bit_cast(f16x8, ip[512 + (t * 2 + 1) * 64]);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xh0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xl0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xh1, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xl1, lo, 0, 0, 0);
    f32x4 hi;
#pragma unroll
    for (int r = 0; r < 4; ++r) hi[r] = lo[r] * F2_DOWN;
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh0, hi, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh1, hi, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.t[t][r] = __builtin_fmaf(hi[r], in.inv, acc.t[t][r]);
  }
}

// diagnostic (-DFE_VBS_WATCHDOG): every spin of the channel-phased kernel gives up after ~0.2 s and leaves its number in g_vbs_dog
#ifdef FE_VBS_WATCHDOG
__device__ int g_vbs_dog[64];
#define VBS_SPIN(id, cond, sl) { long _n = 0; while (cond) { __builtin_amdgcn_s_sleep(sl); if (++_n > 3000000) { if (lane_id() == 0) { atomicAdd(&g_vbs_dog[id], 1); if (blockIdx.x == 0 && g_vbs_dog[48 + wave_id()] == 0) g_vbs_dog[48 + wave_id()] = (id) * 1000000 + _mark; } break; } } }
#define VBS_MARK(x) _mark = (x)
#else
// release builds (round 6): every spin of the channel-phased kernel is BOUNDED -- a hand-off flag that never comes (a bug in
// the READY / TSEQ / ring protocols) ends the wait after ~0.3 s and counts in g_vbs_timeouts instead of hanging the GPU; the launch then
// finishes with wrong numbers, fastegnn_spin_timeouts() (a synchronising query for tests and post-mortems) says so.  Costs one scalar add
// and compare per turn of a spin that practically never turns.
#define VBS_SPIN(id, cond, sl) { int _n = 0; while (cond) { __builtin_amdgcn_s_sleep(sl); if (++_n > VBS_SPIN_LIMIT) { if (lane_id() == 0) atomicAdd(&g_vbs_timeouts, 1); break; } } }
#define VBS_MARK(x)
#endif
constexpr int VBS_SPIN_LIMIT = 1 << 23;
__device__ int g_vbs_timeouts = 0;
template <bool ATT>
__global__ __launch_bounds__(64 * VB_WAVES) void virt_bwd_cs_kernel(VirtCsArgs A) {
  constexpr int SM = GM_F16;
  typedef typename OperandOf<SM>::type SOp;
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef FE_VBS_WATCHDOG
  int _mark = 0;
#endif
#ifdef FE_ISA_CONST   // assembly-only builds of tools/isa_budget_bwd.py: the layer flags as a constant, one graph
  A.f.flags = FE_ISA_CONST; A.f.B = 1;
#endif
  const VirtArgs &a = A.f;
  const int C = a.C;
  constexpr int RMS = rm_lds_bytes<SM>();
  char *rmimg = reinterpret_cast<char *>(lds);                      // V2 | WXV0 | WXX0 (row-major f16x2: product and transpose)
  unsigned *w3 = reinterpret_cast<unsigned *>(lds) + 3 * (RMS / 4);   // W3c[c]^T of the channel in flight (slot c & 1) and of the next
  float *vec = lds + 3 * (RMS / 4) + 2 * VBS_W3_WORDS;
  float *racc0 = vec + VV_COUNT * H;                                // [nbank][5][64]
  float *gBc_l = racc0 + A.nbank * VB_RACC;                         // [2][64]: pool rows of graph `cur`, by channel parity
  float *gZ_l = gBc_l + 2 * H;                                      // [2][4]
  float *ringA = gZ_l + 8;
  float *ringB = ringA + A.ringA * VBS_SLOT_A;
  int *ctrl = reinterpret_cast<int *>(ringB + A.ringB * VB_SLOT_B);
  const int l = lane_id(), j = l & 15, q = l >> 4, wv = wave_id();
  // one wave copies the image of channel c into its stage slot (16 KB: sixteen 16-byte pieces per lane)
  auto stage_w3 = [&](int ph) {   // image of channel ph % C into the slot of phase ph
    const int c = ph % C;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(wpack_x3(a.wpack, C, img_w3ct(C, c)));
    u32x4 *dst = reinterpret_cast<u32x4 *>(w3 + (ph & 1) * VBS_W3_WORDS);
    u32x4 tmp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tmp[i] = src[l + 64 * i];
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[l + 64 * i] = tmp[i];
  };
  {
    const char *src = wpack_rm(a.wpack, C, RM_F16);
    for (int i = threadIdx.x; i < 3 * (RMS / 16); i += blockDim.x) {
      const int im = i / (RMS / 16), k = i % (RMS / 16);
      reinterpret_cast<u32x4 *>(rmimg + im * RMS)[k] = reinterpret_cast<const u32x4 *>(src + (size_t)im * RM_BYTES)[k];
    }
  }
  virt_load_vecs(vec, a);
  for (int i = threadIdx.x; i < A.nbank * VB_RACC + 2 * H + 8; i += blockDim.x) racc0[i] = 0.f;
  if (threadIdx.x < VBSC_CTRL) ctrl[threadIdx.x] = 0;
  const int ntiles = (a.N + 15) >> 4;
  const int t_lo = (int)((long)blockIdx.x * ntiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const int nt = t_hi - t_lo;
  // blocks of EQUAL size near A.block (24.4 tiles per workgroup at cfg4 must not become a block of 24 and one of ONE tile: a phase of one
  // unit keeps two producers busy, not five -- measured: 4.33 instead of 3.93 ms per step)
  const int nblk = max(1, (nt + A.block / 2) / A.block);
  const int BT = (nt + nblk - 1) / nblk;
  const int nphase = nblk * C;
  if (wv == 0) stage_w3(0);
  if (wv == 1 && nphase > 1) stage_w3(1);
  __syncthreads();
  if (threadIdx.x < 2 && (int)threadIdx.x < nphase) ctrl[VBSC_READY + threadIdx.x] = threadIdx.x + 1;   // READY[ph % NPH] == ph + 1: phase ph may run
  __syncthreads();
  // -DFE_VBS_5P=1 (default): wave 6 is a third consumer that takes the dW3c contraction off wave 7 -- with six producers wave 7's two
  // contractions from two rings paced the kernel (virt_bwd 4.41 ms per step; 3.47 with the dW3c products skipped: gpurun_out/cs3)
  constexpr bool FIVEP = FE_VBS_5P != 0;
  const bool consumer = wv == VB_CONS_X || wv == VB_CONS_XX || (FIVEP && wv == VB_CONS_V2);
  const int total = nt * C;                       // units = tickets per ring
  const int cur = a.batch[t_lo * 16];             // graph whose pools this workgroup accumulates in LDS
  const bool tanh_on = a.flags & FASTEGNN_F_TANH;

  if (consumer) {
    __builtin_amdgcn_s_setprio(3);
    const bool wx = wv == VB_CONS_X;              // wave 3: X + XX (ring A);  wave 7: V2 (ring B) + W (ring A)
    auto scp = [&](int slot, int blk, int e4) {
      char *b = reinterpret_cast<char *>(A.cons_scratch + ((size_t)blockIdx.x * 5 + slot) * IMG) + (size_t)((blk * 4 + e4) * 64 * 16);
      asm volatile("" : "+s"(b));
      return reinterpret_cast<f32x4 *>(b + (unsigned)l * 16u);
    };
    // acc (in units of the streams' scales) -> true units, added to the running sum of scratch tile `slot`; last: out as a [o][k] slab
    auto flush_one = [&](WgAcc32 &acc, float ig, float it, int slot, bool have, bool last, float *slab_dst) {
#pragma unroll
      for (int bo = 0; bo < 2; ++bo)
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (acc.c[bo][bk][4 * e4 + r] * ig) * it;
            f32x4 *d = scp(slot, bo * 2 + bk, e4);
            if (have) v += *d;
            if (!last) {
              *d = v;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) slab_dst[(32 * bo + 8 * e4 + 4 * (l >> 5) + r) * H + 32 * bk + (l & 31)] = v[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.c[bo][bk][4 * e4 + r] = 0.f;
          }
    };
    auto bias_out = [&](double (&bs)[2], size_t sl) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        double s0 = bs[b];
        s0 += __shfl_xor(s0, 32);
        if (l < 32) A.slab_b[sl * H + 32 * b + l] = (float)s0;
      }
    };
    auto rowsum8 = [](const float (&x)[2][8], int b) {
      return (double)(((x[b][0] + x[b][1]) + (x[b][2] + x[b][3])) + ((x[b][4] + x[b][5]) + (x[b][6] + x[b][7])));
    };
    WgAcc32 acc0, acc1;      // wave 3: X, XX;  wave 7: V2, W
    wg32_zero(acc0);
    wg32_zero(acc1);
    const int *filledA = ctrl + VBSC_FILLED, *filledB = ctrl + VBSC_FILLED + VB_MAXRING;
    if (wx) {
      WgScale sG0{0}, sG1{0}, sT{0};
      double bs0[2] = {0., 0.}, bs1[2] = {0., 0.};
      bool flushed = false;
      int since = 0;
      int *drained = ctrl + VBSC_DRAINED;
      for (int done = 0; done < total; ++done) {
        VBS_MARK(done);
        const int s0 = done % A.ringA, r0w = done / A.ringA;
        VBS_SPIN(1, vb_ld(&filledA[s0]) != r0w + 1, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const float *g0 = ringA + s0 * VBS_SLOT_A;
        float xt[2][8], xg0[2][8], xg1[2][8];
        wg32_read<VB_RS>(g0 + 2 * VB_TILE, xt);
        wg32_read<VB_RS>(g0, xg0);
        wg32_read<VB_RS>(g0 + VB_TILE, xg1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (l == 0) vb_st(&drained[s0], r0w + 1);
        const float fT = sT.update_lazy(xt);
        const float f0 = sG0.update_lazy(xg0) * fT, f1 = sG1.update_lazy(xg1) * fT;
        if (f0 != 1.f) wg32_scale_acc(acc0, f0);
        if (f1 != 1.f) wg32_scale_acc(acc1, f1);
        const WgOp32 T = wg32_split(xt, sT.scale());
        {
          const WgOp32 G = wg32_split(xg0, sG0.scale());
          bs0[0] += rowsum8(xg0, 0); bs0[1] += rowsum8(xg0, 1);
          wg32_mma(acc0, G, T);
        }
        {
          const WgOp32 G = wg32_split(xg1, sG1.scale());
          bs1[0] += rowsum8(xg1, 0); bs1[1] += rowsum8(xg1, 1);
          wg32_mma(acc1, G, T);
        }
        if (++since >= VB_FLUSH && done + 1 < total) {
          flush_one(acc0, sG0.inv(), sT.inv(), 0, flushed, false, nullptr);
          flush_one(acc1, sG1.inv(), sT.inv(), 1, flushed, false, nullptr);
          flushed = true;
          since = 0;
        }
      }
      const size_t slx = (size_t)A.slab_x + blockIdx.x, slX = (size_t)A.slab_X + blockIdx.x;
      flush_one(acc0, sG0.inv(), sT.inv(), 0, flushed, true, A.slab + slx * IMG);
      flush_one(acc1, sG1.inv(), sT.inv(), 1, flushed, true, A.slab + slX * IMG);
      bias_out(bs0, slx);
      bias_out(bs1, slX);
    } else {
      WgScale sGv{0}, sTv{0}, sGw{0}, sTw{0};      // streams: g_vp, t (ring B);  g_np, v (ring A)
      double bsv[2] = {0., 0.};
      bool flushed_v = false;
      int since_v = 0, doneA = 0, doneB = 0;
      int w_ch = -1;                               // channel whose products acc1 holds
      int w_cnt[2] = {0, 0};                       // tickets of the channel of each parity contracted so far
      bool w_have[2] = {false, false};             // the parity's scratch tile holds a partial sum
      int since_w = 0;
      int *drainedA = ctrl + VBSC_DRAINED + VB_MAXRING, *drainedB = ctrl + VBSC_DRAINED + 2 * VB_MAXRING;
#ifdef FE_VBS_SWAP   // measured alternative: the light (g_vp, t) contraction beside producer 2 on SIMD 2, (g_np, v) beside wave 3 on SIMD 3
      const bool doV2 = FIVEP ? wv == VB_CONS_V2 : wv == VB_CONS_XX, doW = wv == VB_CONS_XX;
#else
      const bool doV2 = wv == VB_CONS_XX, doW = FIVEP ? wv == VB_CONS_V2 : wv == VB_CONS_XX;
#endif
      // acc (+ the parity's scratch tile) -> the channel's running sum over the blocks, ACCUMULATOR order, plain read-modify-write
      // (the slab is this wave's own; the reduction kernel reads that order: WgJob::acc32)
      auto flush_w = [&](WgAcc32 &acc, float ig, float it, int slot, bool have, float *dst, bool add) {
#pragma unroll
        for (int bo = 0; bo < 2; ++bo)
#pragma unroll
          for (int bk = 0; bk < 2; ++bk)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              f32x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (acc.c[bo][bk][4 * e4 + r] * ig) * it;
              if (have) v += *scp(slot, bo * 2 + bk, e4);
              f32x4 *d = reinterpret_cast<f32x4 *>(dst + (size_t)(((bo * 2 + bk) * 4 + e4) * 64 * 4)) + l;
              if (add) v += *d;
              *d = v;
#pragma unroll
              for (int r = 0; r < 4; ++r) acc.c[bo][bk][4 * e4 + r] = 0.f;
            }
      };
      // acc1 -> the running sum of its channel (scratch tile 3 + parity)
      int w_par = 0;                               // phase parity of what acc1 holds
      auto park_w = [&]() {
        if (w_ch < 0) return;
        flush_one(acc1, sGw.inv(), sTw.inv(), 3 + w_par, w_have[w_par], false, nullptr);
        w_have[w_par] = true;
      };
#ifdef FE_VBS_WATCHDOG
      long dog = 0;
#endif
      int idle = 0;   // (release: bounded like VBS_SPIN)
      while ((doW && doneA < total) || (doV2 && doneB < total)) {
        bool did = false;
        if (doV2 && doneB < total) {
          const int s0 = doneB % A.ringB, r0w = doneB / A.ringB;
          if (vb_ld(&filledB[s0]) == r0w + 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            const float *g0 = ringB + s0 * VB_SLOT_B;
            float xg[2][8], xt[2][8];
            wg32_read<VB_RS>(g0, xg);
            wg32_read<VB_RS>(g0 + VB_TILE, xt);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (l == 0) vb_st(&drainedB[s0], r0w + 1);
            const float f = sGv.update_lazy(xg) * sTv.update_lazy(xt);
            if (f != 1.f) wg32_scale_acc(acc0, f);
            const WgOp32 G = wg32_split(xg, sGv.scale()), T = wg32_split(xt, sTv.scale());
            bsv[0] += rowsum8(xg, 0); bsv[1] += rowsum8(xg, 1);
            wg32_mma(acc0, G, T);
            ++doneB;
            if (++since_v >= VB_FLUSH && doneB < total) {
              flush_one(acc0, sGv.inv(), sTv.inv(), 2, flushed_v, false, nullptr);
              flushed_v = true;
              since_v = 0;
            }
            did = true;
          }
        }
        if (doW && doneA < total) {
          const int s0 = doneA % A.ringA, r0w = doneA / A.ringA;
          if (vb_ld(&filledA[s0]) == r0w + 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            const float *g0 = ringA + s0 * VBS_SLOT_A;
            const int chw = __builtin_amdgcn_readfirstlane(ctrl[VBSC_SLOTCH + s0]);
            const int ch = chw & 0xff, nb_ph = (chw >> 8) & 0xfff, first_blk = chw >> 20;   // channel | tiles of its phase | block
            const int par = (first_blk * C + ch) & 1;     // PHASE parity: neighbouring phases differ in it (channels need not: odd C)
            float xg[2][8], xt[2][8];
            wg32_read<VB_RS>(g0 + 3 * VB_TILE, xg);   // g_np
            wg32_read<VB_RS>(g0 + 2 * VB_TILE, xt);   // v
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (l == 0) vb_st(&drainedA[s0], r0w + 1);
            if (ch != w_ch || par != w_par) {          // a ticket of the other phase: park what acc1 holds
              park_w();
              w_ch = ch;
              w_par = par;
              since_w = 0;
            }
#ifndef VBS_DIAG_NO_W   // diagnostic: wave 7 only drains its ring-A tickets (results are wrong without the contraction)
            const float f = sGw.update_lazy(xg) * sTw.update_lazy(xt);
            if (f != 1.f) wg32_scale_acc(acc1, f);
            const WgOp32 G = wg32_split(xg, sGw.scale()), T = wg32_split(xt, sTw.scale());
            wg32_mma(acc1, G, T);
#endif
            ++doneA;
            if (++w_cnt[par] == nb_ph) {               // the phase is complete: into the channel's running sum of this workgroup
              flush_w(acc1, sGw.inv(), sTw.inv(), 3 + par, w_have[par], A.w_slab + ((size_t)ch * gridDim.x + blockIdx.x) * IMG, first_blk != 0);
              w_have[par] = false;
              w_cnt[par] = 0;
              w_ch = -1;
            } else if (++since_w >= VB_FLUSH) {        // cancelling sums: the accumulator leaves the registers every 768 rows
              park_w();
              since_w = 0;
            }
            did = true;
          }
        }
        if (did) idle = 0;
        if (!did) {
          __builtin_amdgcn_s_sleep(1);
          if (++idle > VBS_SPIN_LIMIT) { if (l == 0) atomicAdd(&g_vbs_timeouts, 1); break; }
#ifdef FE_VBS_WATCHDOG
          if (++dog > 6000000) { if (l == 0) { atomicAdd(&g_vbs_dog[6], 1); if (blockIdx.x == 0) { g_vbs_dog[38] = doneA; g_vbs_dog[39] = doneB; g_vbs_dog[40] = total; g_vbs_dog[41] = ctrl[VBSC_UNIT]; g_vbs_dog[42] = ctrl[VBSC_HEAD]; g_vbs_dog[43] = ctrl[VBSC_HEAD + 1]; for (int _k = 0; _k < 8; ++_k) { g_vbs_dog[8 + _k] = ctrl[VBSC_READY + _k]; g_vbs_dog[16 + _k] = ctrl[VBSC_DONE + _k]; } } } break; }
#endif
        }
      }
      if (doV2) {
        const size_t sl = (size_t)A.slab_v2 + blockIdx.x;
        flush_one(acc0, sGv.inv(), sTv.inv(), 2, flushed_v, true, A.slab + sl * IMG);
        bias_out(bsv, sl);
      }
    }
  } else {
    // ---------------------------------------------------------------------------------------------------------
    // producers
    // ---------------------------------------------------------------------------------------------------------
    const float invC = 1.0f / (float)C;
    const float attb0 = ATT ? a.attb[0] : 0.f;
    float *racc = racc0 + ((wv > 3 ? wv - 1 : wv) % A.nbank) * VB_RACC;
    auto mm = [&](int which, const SOp &op, Vec &acc) { gemm_rm<SM, false>(rmimg + which * RMS, op, acc); };
    auto mmT = [&](int which, const Vec &g, Vec &acc) {
      const auto op = make_grad_operand<SM>(g);
      gemm_rm_g<SM, true>(rmimg + which * RMS, op, acc);
    };
    int pend_tile = -1, pend_c = 0;   // the unit whose rows this wave stored last: not yet handed on
    VB2_T0()   // (phase stamps of the producers, -DFE_STAMP builds only: tools/gpu_stamp_vbs.py)
    auto take = [&]() {
      int u_ = 0;
      if (l == 0) u_ = atomicAdd(&ctrl[VBSC_UNIT], 1);
      return __builtin_amdgcn_readfirstlane(u_);
    };
    auto unit_of = [&](int u_, int &blk_, int &nb_, int &c_, int &tile_) {
      const int per_blk = BT * C;
      blk_ = min(u_ / per_blk, nblk - 1);
      nb_ = min(BT, nt - blk_ * BT);
      const int ru = u_ - blk_ * per_blk;
      c_ = ru / nb_;
      tile_ = t_lo + blk_ * BT + (ru - c_ * nb_);
    };
    for (;;) {
      const int u = take();
      if (u >= total) {
        if (pend_tile >= 0) {   // (units of other waves may still wait for this wave's last rows)
          __builtin_amdgcn_s_waitcnt(0x0f70);
          asm volatile("" ::: "memory");
          if (l == 0) __hip_atomic_store(&ctrl[VBSC_TSEQ + pend_tile], pend_c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        break;
      }
      // unit -> (block, channel, tile): blocks of VBS_BLOCK tiles (the last one shorter), channel-major inside a block
      int blk, nb, c, tile;                                         // nb: tiles of this block
      unit_of(u, blk, nb, c, tile);
      const int ph = blk * C + c;                                   // phase: stage slot / pool rows ph & 1, flags ph % VBS_NPH
      VBS_MARK(u);
      if (pend_tile >= 0 && (vb_ld(&ctrl[VBSC_READY + ph % VBS_NPH]) != ph + 1 || vb_ld(&ctrl[VBSC_TSEQ + (tile - t_lo)]) < c)) {
        // about to wait (rare): hand this wave's last rows on first -- a wave that waits must not hold back what others wait for
        __builtin_amdgcn_s_waitcnt(0x0f70);
        asm volatile("" ::: "memory");
        if (l == 0) __hip_atomic_store(&ctrl[VBSC_TSEQ + pend_tile], pend_c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        pend_tile = -1;
      }
      VBS_SPIN(2, vb_ld(&ctrl[VBSC_READY + ph % VBS_NPH]) != ph + 1, 2);   // W3c[c]^T is in slot ph & 1, the parity's pool rows are clear
      // the tile's g_A / g_x rows were last written by whichever wave of THIS workgroup held (c - 1, tile): workgroup scope -- the waves
      // of a workgroup share their CU's vector L1, so acquire / release cost a wait, no cache maintenance (agent scope writes the L2
      // back and invalidates the L1 on this multi-XCD part: measured 4.1 instead of 0.7 ms per launch)
      // TSEQ[tile] = c once the rows of (c - 1, tile) are in memory -- published by their writer lazily, behind its NEXT unit's first
      // loads (below), so nobody ever waits for a store: this spin practically never turns
      VBS_SPIN(7, vb_ld(&ctrl[VBSC_TSEQ + (tile - t_lo)]) < c, 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      VB2_T(0)   // unit ticket, waits for the phase's image / the tile's rows
      const int n0 = tile * 16, nend = min(a.N, n0 + 16);
      // (one graph in the batch -- the frames this form is for: no batch lookups at the head of the dependent-load chain)
      const int b0 = a.B == 1 ? 0 : a.batch[n0], b1 = a.B == 1 ? 0 : a.batch[nend - 1];
      const int pmode = b0 != b1 ? 2 : (b0 == cur ? 0 : 1);
      const int n = n0 + j;
      const bool valid = n < nend;
      const int nc = valid ? n : nend - 1;
      const int b = a.B == 1 ? 0 : a.batch[nc];
      const unsigned offB = (unsigned)b * C * H + 4u * q + (unsigned)c * H;   // [B,C,64] arrays
      const unsigned offN = (unsigned)nc * H + 4u * q;                        // [N,64] arrays
      const unsigned offZ = (unsigned)b * 3u * C;
      float gxn[3], xi[3], gx[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gxn[k] = valid ? A.g_x_out[(size_t)nc * 3 + k] : 0.f;
        xi[k] = a.x[(size_t)nc * 3 + k];
        // the tile's g_x accumulates through memory; the direct path x' = x + ... is counted once, with channel 0
        gx[k] = c == 0 ? gxn[k] : (valid ? A.g_x[(size_t)nc * 3 + k] : 0.f);
      }
      float vd[3], gpX[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vd[k] = (a.Z + (k * C + c))[offZ] - xi[k];
        gpX[k] = valid ? (A.g_poolX + (k * C + c))[offZ] : 0.f;
      }
      const float vr = sqrt_f(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]);
      Vec vp = vload_vec(vec + VV_C2 * H, q);
      Vec d_pre = vload_u(a.A, offN);
      vadd(d_pre, vload_u(a.Bc, offB));
      vaxpy(d_pre, vr, vload_vec(vec + VV_WVR * H, q));
      const Vec t = vsilu_keep_d(d_pre FE_ACT(a));        // d_pre <- silu'(pre)
      if (pend_tile >= 0) {
        // every load of this unit's head has returned, and the vector memory counter runs in order: the g_A / g_x rows the PREVIOUS
        // unit of this wave stored are in memory.  Its tile may now be taken up in the next channel.
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        asm volatile("" ::: "memory");
        if (l == 0) __hip_atomic_store(&ctrl[VBSC_TSEQ + pend_tile], pend_c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
      }
      VB2_T(1)   // head loads (rows of A / Bc, coordinates), pre-activation, SiLU 1 (+ the hand-on of the previous unit's rows)
      mm(0, make_operand<SM>(t), vp);
      VB2_T(2)   // operand split + V2 product
      const Vec v0 = vsilu_keep_d(vp FE_ACT(a));          // vp <- silu'(vp)
      float att = 1.f;
      Vec v = v0;
      if constexpr (ATT) {
        att = sigmoid_f(vdot(v0, vload_vec(vec + VV_ATT * H, q)) + attb0);
        v = vscale(v0, att);
      }
      VB2_T(3)   // SiLU 2
      float sx, sX;
      Vec g_ux, g_uX;
      {
        Vec uxp = vload_vec(vec + VV_BXV0 * H, q), uXp = vload_vec(vec + VV_BXX0 * H, q);
        {
          const SOp vs = make_operand<SM>(v);
          mm(1, vs, uxp);
          mm(2, vs, uXp);
        }
        VB2_T(4)   // operand split of v + the two head products
        {
          Vec ux = vsilu_keep_d(uxp FE_ACT(a));
          const float sr = vdot(ux, vload_vec(vec + VV_WXV2 * H, q));
          sx = tanh_on ? tanh_f(sr) : sr;
          float g_sx = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) g_sx -= vd[k] * invC * gxn[k];
          const float g_sr = tanh_on ? g_sx * (1.f - sx * sx) : g_sx;
          vb_accum_items(racc + 0 * H, vscale(ux, g_sr), j, q);
          g_ux = vmul(vscale(vload_vec(vec + VV_WXV2 * H, q), g_sr), uxp);
        }
        VB2_T(5)   // head x: SiLU, head dot, rank-1 sum (DPP + LDS atomic), g_ux
        {
          Vec uX = vsilu_keep_d(uXp FE_ACT(a));
          const float sr = vdot(uX, vload_vec(vec + VV_WXX2 * H, q));
          sX = tanh_on ? tanh_f(sr) : sr;
          float g_sX = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) g_sX += vd[k] * gpX[k];
          const float g_sr = tanh_on ? g_sX * (1.f - sX * sX) : g_sX;
          vb_accum_items(racc + 1 * H, vscale(uX, g_sr), j, q);
          g_uX = vmul(vscale(vload_vec(vec + VV_WXX2 * H, q), g_sr), uXp);
        }
      }
      VB2_T(6)   // head X: the same
      // Gv = g_poolV[b, c] + W3c[c]^T g_np is formed in registers; the g_np row also rides to wave 7 in the ring slot
      const Vec gnp = vb_mask(vload_u(A.g_np, offN), valid);
      {   // (g_ux, v), (g_uX, v), (g_np, v) to waves 3 and 7: one slot of ring A, free once both have drained it
        int tk = 0;
        if (l == 0) tk = atomicAdd(&ctrl[VBSC_HEAD + 0], 1);
        tk = __builtin_amdgcn_readfirstlane(tk);
        const int sl = tk % A.ringA, round = tk / A.ringA;
        VBS_SPIN(4, vb_ld(&ctrl[VBSC_DRAINED + 0 * VB_MAXRING + sl]) != round, 2);
        VBS_SPIN(5, vb_ld(&ctrl[VBSC_DRAINED + 1 * VB_MAXRING + sl]) != round, 2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        float *slot = ringA + sl * VBS_SLOT_A;
        vb_tile_store(slot, j, q, g_ux);
        vb_tile_store(slot + VB_TILE, j, q, g_uX);
        vb_tile_store(slot + 2 * VB_TILE, j, q, valid ? v : vzero());
        vb_tile_store(slot + 3 * VB_TILE, j, q, gnp);
        if (l == 0) ctrl[VBSC_SLOTCH + sl] = c | (nb << 8) | (blk << 20);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (l == 0) vb_st(&ctrl[VBSC_FILLED + sl], round + 1);
        asm volatile("" ::: "memory");
      }
      VB2_T(7)   // g_np row, publish (g_ux, g_uX, v, g_np) to ring A (incl. the wait for a drained slot)
      Vec g_v = vb_mask(vload_u(A.g_poolV, offB), valid);
      gemm64_f2_scaled(w3 + (ph & 1) * VBS_W3_WORDS, vsplit2_scaled(gnp), g_v);
      mmT(1, g_ux, g_v);
      mmT(2, g_uX, g_v);
      VB2_T(8)   // three scaled gradient splits + W3c^T g_np and the two transposed head products
      float g_vd[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) g_vd[k] = -sx * invC * gxn[k] + sX * gpX[k];
      Vec g_v0 = g_v;
      if constexpr (ATT) {
        const float g_a = vdot(g_v, v0);
        const float g_z = g_a * att * (1.f - att);
        vb_accum_items(racc + 3 * H, vscale(v0, g_z), j, q);
        const float sz = jsum(q == 0 ? g_z : 0.f);
        if (l == 0) atomicAdd(&racc[4 * H], sz);
        g_v0 = vscale(g_v, att);
        vaxpy(g_v0, g_z, vload_vec(vec + VV_ATT * H, q));
      }
      Vec g_t = vzero();
      Vec ga = vzero();
      {
        const Vec g_vp = vmul(g_v0, vp);
        {   // (g_vp, t) to wave 7
          int tk = 0;
          if (l == 0) tk = atomicAdd(&ctrl[VBSC_HEAD + 1], 1);
          tk = __builtin_amdgcn_readfirstlane(tk);
          const int sl = tk % A.ringB, round = tk / A.ringB;
          VBS_SPIN(3, vb_ld(&ctrl[VBSC_DRAINED + 2 * VB_MAXRING + sl]) != round, 2);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
          float *slot = ringB + sl * VB_SLOT_B;
          vb_tile_store(slot, j, q, g_vp);
          vb_tile_store(slot + VB_TILE, j, q, t);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          if (l == 0) vb_st(&ctrl[VBSC_FILLED + VB_MAXRING + sl], round + 1);
          asm volatile("" ::: "memory");
        }
        if (c > 0) ga = vload_u(A.g_A, offN);     // requested here, consumed after the product: the tile's running g_A
        mmT(0, g_vp, g_t);
      }
      VB2_T(9)   // g_vp, publish (g_vp, t) to ring B, scaled split + V2^T product
      const Vec g_pre = vmul(g_t, d_pre);
      vadd(ga, g_pre);
      if (valid) vstore_u(A.g_A, offN, ga);
      vb_accum_items(racc + 2 * H, vscale(g_pre, vr), j, q);
      const float g_vr = vdot(g_pre, vload_vec(vec + VV_WVR * H, q));
      const float ivr = vr > 0.f ? g_vr * rcp_f(vr) : 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        g_vd[k] += ivr * vd[k];
        gx[k] -= g_vd[k];
      }
      if (valid && q == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) A.g_x[(size_t)n * 3 + k] = gx[k];
      }
      // pools over the nodes of the tile: g_Bc[b,c,:] += g_pre, g_Zp[b,:,c] += g_vd
      if (pmode < 2) {
        float pz[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pz[k] = jsum((valid && q == 0) ? g_vd[k] : 0.f);
        vb_accum_items(pmode == 0 ? gBc_l + (ph & 1) * H : A.g_Bc + ((size_t)b0 * C + c) * H, g_pre, j, q);
        if (l < 3) {
          const float pzl = l == 0 ? pz[0] : (l == 1 ? pz[1] : pz[2]);
          if (pmode == 0) atomicAdd(&gZ_l[(ph & 1) * 4 + l], pzl);
          else atomicAdd(&A.g_Zp[((size_t)b0 * 3 + l) * C + c], pzl);
        }
      } else {
        if (valid && q == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) atomicAdd(&A.g_Zp[((size_t)b * 3 + k) * C + c], g_vd[k]);
        }
        if (valid) {
#pragma unroll
          for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              atomicAdd(&A.g_Bc[((size_t)b * C + c) * H + 16 * t2 + 4 * q + r], g_pre.t[t2][r]);
        }
      }
      // unit done.  Its pool atomics (LDS) are out before the channel's counter moves; its g_A / g_x rows are handed on lazily (TSEQ)
      pend_tile = tile - t_lo;
      pend_c = c;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      int dn = 0;
      if (l == 0) dn = atomicAdd(&ctrl[VBSC_DONE + ph % VBS_NPH], 1);
      dn = __builtin_amdgcn_readfirstlane(dn);
      if (dn == nb - 1) {
        // this wave closes channel c: the parity's pool rows go out and are cleared, the stage slot takes channel c + 2
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const float pb = gBc_l[(ph & 1) * H + l];
        atomicAdd(&A.g_Bc[((size_t)cur * C + c) * H + l], pb);
        gBc_l[(ph & 1) * H + l] = 0.f;
        if (l < 3) {
          atomicAdd(&A.g_Zp[((size_t)cur * 3 + l) * C + c], gZ_l[(ph & 1) * 4 + l]);
          gZ_l[(ph & 1) * 4 + l] = 0.f;
        }
        if (l == 0) ctrl[VBSC_DONE + ph % VBS_NPH] = 0;   // (the flag slot is reused VBS_NPH phases later)
        if (ph + 2 < nphase) {
          stage_w3(ph + 2);
          // (a RELEASE store at workgroup scope between compiler barriers: as a relaxed store behind a fence the compiler sank it out
          // of the unit loop -- legal for a relaxed atomic, fatal here: this very wave goes on to spin on the flag of a later channel
          // while the others wait for this one.  Found with the spin watchdog of -DFE_VBS_WATCHDOG, tools/gpu_vbs_dog.py)
          asm volatile("" ::: "memory");
          if (l == 0) __hip_atomic_store(&ctrl[VBSC_READY + (ph + 2) % VBS_NPH], ph + 3, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
        }
      }
      VB2_T(10)   // g_pre: g_A read-modify-write, w_vr rank-1 sum, g_x, pools, phase counter (+ the next image when this unit closes a phase)
    }
    VB2_TEND(0)
  }
  VB2_T0()
  __syncthreads();
  VB2_T(11)   // wait for the rest of the workgroup
  VB2_TEND(consumer ? 12 : 0)
  if (threadIdx.x < H) {
    const int o = threadIdx.x;
    float s5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int bk = 0; bk < A.nbank; ++bk) {
      const float *r = racc0 + bk * VB_RACC;
#pragma unroll
      for (int k = 0; k < 4; ++k) s5[k] += r[k * H + o];
      s5[4] += r[4 * H];
    }
    atomicAdd(&A.d_wxv2[o], s5[0]);
    atomicAdd(&A.d_wxx2[o], s5[1]);
    atomicAdd(&A.d_wvr[(size_t)o * A.ld_v0], s5[2]);
    if constexpr (ATT) {
      atomicAdd(&A.d_attw[o], s5[3]);
      if (o == 0) atomicAdd(A.d_attb, s5[4]);
    }
  }
}

// does the channel-phased form run this layer's B4?  (default build, fp32-grade SiLU-or-generic activations, FastEGNN wiring, enough
// tiles per workgroup; FASTEGNN_VIRT_CS=0 in the environment keeps the tile-major form -- an A/B switch)
// workgroups of the channel-phased form: one per CU while every workgroup keeps >= VBS_MIN_TILES tiles (the g_A / g_x rows of a tile
// pass from unit (c, tile) to unit (c + 1, tile) through memory: the two must never be in flight together)
static int virt_cs_grid(long N) {
  const long g = ((N + 15) / 16) / VBS_MIN_TILES;
  return (int)(g > 256 ? 256 : g);
}
// the form runs when that fills the chip; FASTEGNN_VIRT_CS_MIN_GRID lowers the bar (tests: the oracle comparisons at 20 000 nodes)
static int virt_cs_min_grid() {
  static const int v = getenv("FASTEGNN_VIRT_CS_MIN_GRID") ? atoi(getenv("FASTEGNN_VIRT_CS_MIN_GRID")) : 256;
  return v < 1 ? 1 : v;
}
bool virt_cs_applies(long N, int C, int flags) {
  if (!(FE_VB_CS && GM_VIRT_BWD == GM_F16 && FE_VB_CONS32)) return false;
  static const bool off = getenv("FASTEGNN_VIRT_CS") && atoi(getenv("FASTEGNN_VIRT_CS")) == 0;
  if (off || (flags & (FASTEGNN_F_BF16 | FASTEGNN_F_RF | FASTEGNN_F_EGNN)) || C < 1 || C > VBS_MAXC) return false;
  const int grid = virt_cs_grid(N);
  if (grid < 1) return false;
  // Upper bound on the tiles per workgroup (default 64): measured (profiles/r05_lever_virt_cs.txt), at 24 tiles per workgroup (cfg4) the
  // two forms tie on time and this one keeps 1.6 GB per launch out of HBM; at 244 (cfg5, walked in blocks of 24) it needs 12.2 instead
  // of 20.3 GB and is 1.6 % slower per step (its units pay a g_np split and a g_A read-modify-write per channel) -- the tile-major
  // form runs there unless FASTEGNN_VIRT_CS_MAX_TILES=256 asks for the smaller footprint.
  static const int max_tiles = getenv("FASTEGNN_VIRT_CS_MAX_TILES") ? atoi(getenv("FASTEGNN_VIRT_CS_MAX_TILES")) : 64;
  const long nt = ((N + 15) / 16 + grid - 1) / grid;
  return grid >= virt_cs_min_grid() && nt <= VBS_MAXTILES && nt <= max_tiles;
}
size_t virt_cs_wg_floats(size_t C) { return (size_t)256 * (5 + C) * IMG; }

// floats of wg_virt: Gv, overwritten row by row with v ([C][N + pad][64]) | parts of g_A and g_x | consumer scratch
size_t virt_pc_wg_floats(size_t N, size_t C) {
  const size_t NGF = (C + VB_GF - 1) / VB_GF;
  return (N + WGV_PAD) * C * H + (size_t)256 * VB_FINE_TILES * (NGF - 1) * 16 * (H + 4) + (size_t)256 * 3 * IMG;
}

// B4c': the (node, channel) part in the channel-phased form -- one kernel, then the fixed-order reduction of its partial slabs
static int virt_backward_channels_cs(const fastegnn_layer_t *L, hipStream_t st, float *wg_gnp) {
  const int C = L->C;
  const bool att = has(L, FASTEGNN_F_ATTENTION);
  float *const *g = L->grads;
  const int grid = virt_cs_grid(L->N);
  FE_REQUIRE(L->g_poolV, "virt_backward: g_poolV null");
  WgradBatch bb(L->wg_slab, st, false, WG_SLABS / 2, WG_SLABS / 2);
  VirtCsArgs A;
  A.f = make_virt_args(L);
  A.g_x_out = L->g_x_out; A.g_poolX = L->g_poolX; A.g_poolV = L->g_poolV; A.g_np = wg_gnp;
  A.g_x = L->g_x; A.g_A = L->g_A; A.g_Bc = L->g_Bc; A.g_Zp = L->g_Zp;
  A.cons_scratch = L->wg_virt;
  A.w_slab = L->wg_virt + (size_t)grid * 5 * IMG;
  A.ld_v0 = 2 * H + 1 + C;
  A.d_wxv2 = g[FASTEGNN_P_CRV2_W]; A.d_wxx2 = g[FASTEGNN_P_CVV2_W];
  A.d_wvr = g[FASTEGNN_P_VIRT0_W] + 2 * H;
  A.d_attw = g[FASTEGNN_P_ATTV_W]; A.d_attb = g[FASTEGNN_P_ATTV_B];
  int rc;
  if ((rc = bb.add_slabs(g[FASTEGNN_P_CRV0_W], H, 0, 1, g[FASTEGNN_P_CRV0_B], grid, &A.slab_x))) return rc;
  if ((rc = bb.add_slabs(g[FASTEGNN_P_CVV0_W], H, 0, 1, g[FASTEGNN_P_CVV0_B], grid, &A.slab_X))) return rc;
  if ((rc = bb.add_slabs(g[FASTEGNN_P_VIRT2_W], H, 0, 1, g[FASTEGNN_P_VIRT2_B], grid, &A.slab_v2))) return rc;
  A.slab = bb.tab.slab; A.slab_b = bb.tab.slab_b;
  // rings: 3 slots of (g_ux | g_uX | v | g_np) + 2 of (g_vp | t) are what the 160 KB leave beside five images; FE_VBS_RING overrides
  static const int ring_want = getenv("FE_VBS_RING") ? atoi(getenv("FE_VBS_RING")) : 32;
  static const int block_want = getenv("FASTEGNN_VIRT_CS_BLOCK") ? atoi(getenv("FASTEGNN_VIRT_CS_BLOCK")) : VBS_BLOCK;
  A.block = block_want < 1 ? 1 : (block_want > 0xfff ? 0xfff : block_want);
  A.ringA = ring_want / 10; A.ringB = ring_want % 10; A.nbank = 1;
  if (A.ringA < 2 || A.ringA > VB_MAXRING) A.ringA = 3;
  if (A.ringB < 2 || A.ringB > VB_MAXRING) A.ringB = 2;
  while (vbs_lds_floats(A.ringA, A.ringB, 1) * sizeof(float) > 160 * 1024 && A.ringB > 2) --A.ringB;
  while (vbs_lds_floats(A.ringA, A.ringB, 1) * sizeof(float) > 160 * 1024 && A.ringA > 2) --A.ringA;
  while (A.nbank < VB_MAXBANK && vbs_lds_floats(A.ringA, A.ringB, A.nbank + 1) * sizeof(float) <= 160 * 1024) ++A.nbank;
  const size_t lds = vbs_lds_floats(A.ringA, A.ringB, A.nbank) * sizeof(float);
  FE_REQUIRE(lds <= 160 * 1024, "virt_backward: LDS budget exceeded");
  {
    ProfScope ps(K_VIRT_BWD, st);
    const dim3 g3(grid), b3(64 * VB_WAVES);
    if (att) hipLaunchKernelGGL((virt_bwd_cs_kernel<true>), g3, b3, lds, st, A);
    else hipLaunchKernelGGL((virt_bwd_cs_kernel<false>), g3, b3, lds, st, A);
  }
  if ((rc = check_launch("virt_bwd_cs_kernel"))) return rc;
  // node_mlp.0 block of channel c (column 2H + k C + c, k = feature of v): one slab set per channel, written by wave 7
  const int ld_n0 = 2 * H + H * C + L->na;
  if ((rc = bb.add_slabs_ext(A.w_slab, g[FASTEGNN_P_NODE0_W], ld_n0, 2 * H, C, grid, C, 1))) return rc;
  return bb.finish();
}

// B4b + B4c: the (node, channel) part -- Gv, then the producer / consumer kernel and its weight gradients
static int virt_backward_channels(const fastegnn_layer_t *L, hipStream_t st, float *wg_gnp) {
  if (virt_cs_applies(L->N, L->C, L->flags)) return virt_backward_channels_cs(L, st, wg_gnp);
  const int N = L->N, C = L->C, ntiles = cdiv(N, 16);
  const bool bf = has(L, FASTEGNN_F_BF16), att = has(L, FASTEGNN_F_ATTENTION), rf = has(L, FASTEGNN_F_RF);
  float *const *g = L->grads;
  const size_t cstride = ((size_t)N + WGV_PAD) * H;
  float *wg_v = L->wg_virt, *Gv = L->wg_virt;   // one array: virt_bwd_gv writes Gv, virt_bwd_pc replaces each row by v
  const int NGF = (C + VB_GF - 1) / VB_GF;
  float *gA_part = Gv + cstride * C, *gx_part = gA_part + (size_t)256 * VB_FINE_TILES * (NGF - 1) * 16 * H;
  int rc;
  if (rf) {   // FastRF: no node_model and no pooled messages -- d/dv is the recomputed heads' part alone
    (void)hipMemsetAsync(Gv, 0, cstride * C * sizeof(float), st);
  } else {
    const int ngroups = cdiv(C, VB_GV_CH);
    int nranges = 256 / ngroups;
    const int max_ranges = cdiv(ntiles, VB_GV_WAVES);   // at least one tile per wave where possible
    if (nranges > max_ranges) nranges = max_ranges;
    if (nranges < 1) nranges = 1;
    VirtGvArgs a{wg_gnp, L->g_poolV, L->wpack, L->batch, Gv, cstride, N, C, ngroups, nranges};
    const size_t lds = (size_t)VB_GV_CH * RM_BYTES;
    ProfScope ps(K_VIRT_BWD_GV, st);
    if (bf) hipLaunchKernelGGL((virt_bwd_gv_kernel<true>), dim3(ngroups * nranges), dim3(64 * VB_GV_WAVES), lds, st, a);
    else hipLaunchKernelGGL((virt_bwd_gv_kernel<false>), dim3(ngroups * nranges), dim3(64 * VB_GV_WAVES), lds, st, a);
  }
  if ((rc = check_launch("virt_bwd_gv_kernel"))) return rc;
  // the three in-kernel weight gradients and the per-channel node_mlp.0 blocks: a batch of their own, upper half of the slabs
  WgradBatch bb(L->wg_slab, st, bf, WG_SLABS / 2, WG_SLABS / 2);
  VirtBwd2Args A;
  A.f = make_virt_args(L);
  A.g_x_out = L->g_x_out; A.g_poolX = L->g_poolX; A.Gv = Gv;
  A.g_x = L->g_x; A.g_A = L->g_A; A.g_Bc = L->g_Bc; A.g_Zp = L->g_Zp;
  A.gA_part = gA_part; A.gx_part = gx_part; A.wg_v = wg_v; A.cstride = cstride;
  A.cons_scratch = gx_part + (size_t)256 * VB_FINE_TILES * (NGF - 1) * 16 * 4;
  A.ld_v0 = 2 * H + 1 + C;
  A.d_wxv2 = g[FASTEGNN_P_CRV2_W]; A.d_wxx2 = g[FASTEGNN_P_CVV2_W];
  A.d_wvr = g[FASTEGNN_P_VIRT0_W] + 2 * H;
  A.d_attw = g[FASTEGNN_P_ATTV_W]; A.d_attb = g[FASTEGNN_P_ATTV_B];
  A.NGF = NGF;
  int grid = ntiles < 256 ? ntiles : 256;   // one workgroup per CU (LDS), each with an equal share of the tiles
  if ((rc = bb.add_slabs(g[FASTEGNN_P_CRV0_W], H, 0, 1, g[FASTEGNN_P_CRV0_B], grid, &A.slab_x))) return rc;
  if ((rc = bb.add_slabs(g[FASTEGNN_P_CVV0_W], H, 0, 1, g[FASTEGNN_P_CVV0_B], grid, &A.slab_X))) return rc;
  if ((rc = bb.add_slabs(g[FASTEGNN_P_VIRT2_W], H, 0, 1, g[FASTEGNN_P_VIRT2_B], grid, &A.slab_v2))) return rc;
  A.slab = bb.tab.slab; A.slab_b = bb.tab.slab_b;
  // rings: as many slots as the 160 KB of LDS leave
  // ring slots: as many as FE_VB_RING asks for and the LDS holds; the f16x2 images leave 28 KB more than the bf16 ones.  Default
  // 5 + 4 (round 4, two repeats on one box, tools/gpu_ab_rings2.sh: 3+3 3.47, 4+4 3.42, 5+4 3.40, 5+5 3.42, 6+4 3.47 ms per step)
  static const int ring_want = getenv("FE_VB_RING") ? atoi(getenv("FE_VB_RING")) : 54;   // two digits: ring A, ring B
  const int imgb = bf ? RM_BYTES : (GM_VIRT_BWD == GM_F16 ? rm_lds_bytes<GM_F16>() : RM_BYTES);
  A.ringA = ring_want / 10; A.ringB = ring_want % 10; A.nbank = 1;
  if (A.ringA < 2 || A.ringA > VB_MAXRING) A.ringA = 3;
  if (A.ringB < 2 || A.ringB > VB_MAXRING) A.ringB = 3;
  while (vb_lds_floats(C, A.ringA, A.ringB, 1, imgb) * sizeof(float) > 160 * 1024 && A.ringB > 2) --A.ringB;
  while (vb_lds_floats(C, A.ringA, A.ringB, 1, imgb) * sizeof(float) > 160 * 1024 && A.ringA > 2) --A.ringA;
  while (A.nbank < VB_MAXBANK && vb_lds_floats(C, A.ringA, A.ringB, A.nbank + 1, imgb) * sizeof(float) <= 160 * 1024) ++A.nbank;
  const size_t lds = vb_lds_floats(C, A.ringA, A.ringB, A.nbank, imgb) * sizeof(float);
  FE_REQUIRE(lds <= 160 * 1024, "virt_backward: LDS budget exceeded");
  {
    ProfScope ps(K_VIRT_BWD, st);
    const dim3 g3(grid), b3(64 * VB_WAVES);
    if (bf) { if (att) hipLaunchKernelGGL((virt_bwd_pc_kernel<true, true>), g3, b3, lds, st, A); else hipLaunchKernelGGL((virt_bwd_pc_kernel<true, false>), g3, b3, lds, st, A); }
    else { if (att) hipLaunchKernelGGL((virt_bwd_pc_kernel<false, true>), g3, b3, lds, st, A); else hipLaunchKernelGGL((virt_bwd_pc_kernel<false, false>), g3, b3, lds, st, A); }
  }
  if ((rc = check_launch("virt_bwd_pc_kernel"))) return rc;
  if (NGF > 1) {
    hipLaunchKernelGGL(virt_bwd_combine_kernel, dim3(grid * VB_FINE_TILES), dim3(256), 0, st, L->g_A, L->g_x, gA_part, gx_part, N, NGF - 1, grid);
    if ((rc = check_launch("virt_bwd_combine_kernel"))) return rc;
  }
  // node_mlp.0 block of channel c: (g_np, v[:, c]) -- one batch slice per channel
  if (!rf) {
    const int ld_n0 = 2 * H + H * C + L->na;
    if ((rc = bb.add(wg_gnp, H, wg_v, H, N, g[FASTEGNN_P_NODE0_W], ld_n0, 2 * H, C, nullptr, C, 0, (long)cstride, 1))) return rc;
  }
  return bb.finish();
}

// One form for the three wirings: FastEGNN; FastRF (FASTEGNN_F_RF: no node_model -- B4a passes g_h through, Gv = 0, no
// node_mlp jobs); the EGNN baseline (FASTEGNN_F_EGNN, C = 0: B4a and the node_mlp jobs only).
int virt_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared) {
  const bool egnn = has(L, FASTEGNN_F_EGNN), rf = has(L, FASTEGNN_F_RF);
  FE_REQUIRE(L->h && L->A && L->x && L->vel && L->aggm && L->npre && L->batch && L->wpack && (!egnn || L->aggx),
             "virt_backward: null saved buffer");
  FE_REQUIRE(L->g_h_out && L->g_x_out && L->g_h && L->g_x && L->g_A && L->g_aggm && L->g_aggx && L->g_svel && L->wg_node &&
                 L->grads && L->wg_slab,
             "virt_backward: null gradient buffer");
  FE_REQUIRE(egnn ? L->C == 0
                  : (L->Bc && L->Z && (rf || L->g_poolV) && L->g_poolX && L->g_Bc && L->g_Zp && L->wg_virt && L->C >= 1 && L->C <= 64),
             "virt_backward: virtual buffers null or virtual_channels outside [1,64] (0 with FASTEGNN_F_EGNN)");
  const int N = L->N, C = L->C;
  if (C > 0) {
    (void)hipMemsetAsync(L->g_Bc, 0, (size_t)L->B * C * H * sizeof(float), st);
    (void)hipMemsetAsync(L->g_Zp, 0, (size_t)L->B * 3 * C * sizeof(float), st);
  }
  if (N == 0) return check_launch("virt_backward(memset)");
  const bool bf = has(L, FASTEGNN_F_BF16);
  float *const *g = L->grads;
  FE_REQUIRE(!has(L, FASTEGNN_F_ATTENTION) || C == 0 || (g[FASTEGNN_P_ATTV_W] && g[FASTEGNN_P_ATTV_B]), "virt_backward: attention grads null");
  FE_REQUIRE(!has(L, FASTEGNN_F_GRAVITY) || L->g_sgrav, "virt_backward: g_sgrav null");
  float *wg_t3 = L->wg_node, *wg_gnp = L->wg_node + (size_t)N * H;
  const int ntiles = cdiv(N, 16);
  int rc;
  {   // B4a
    VirtNodeArgs a{L->g_h_out, L->npre, L->g_x_out, L->vel, L->aggx, L->wpack, wg_t3, wg_gnp, L->g_h, L->g_aggm, L->g_aggx,
                   L->g_svel, L->g_sgrav, N, L->flags, C, {L->gravity[0], L->gravity[1], L->gravity[2]}, L->act_param};
    int grid = cdiv(ntiles, VB_NODE_WAVES);
    if (grid > 256) grid = 256;
    ProfScope ps(K_VIRT_BWD_NODE, st);
    if (bf) hipLaunchKernelGGL((virt_bwd_node_kernel<true>), dim3(grid), dim3(64 * VB_NODE_WAVES), 3 * IMG3 * sizeof(float), st, a);
    else hipLaunchKernelGGL((virt_bwd_node_kernel<false>), dim3(grid), dim3(64 * VB_NODE_WAVES), 3 * IMG3 * sizeof(float), st, a);
  }
  if ((rc = check_launch("virt_bwd_node_kernel"))) return rc;
  if (C == 0) {   // no virtual nodes: the coordinate gradient passes through, nothing reaches A
    (void)hipMemcpyAsync(L->g_x, L->g_x_out, (size_t)N * 3 * sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemsetAsync(L->g_A, 0, (size_t)N * H * sizeof(float), st);
    if ((rc = check_launch("virt_backward(C = 0)"))) return rc;
  } else if ((rc = virt_backward_channels(L, st, wg_gnp))) {
    return rc;
  }
  if (rf) return FASTEGNN_OK;   // no node_mlp
  WgradBatch local(L->wg_slab, st);
  WgradBatch &wb = shared ? *shared : local;
  wb.round = bf;
  const int ld_n0 = 2 * H + H * C + L->na;
  // node_mlp.2
  if ((rc = wb.add(L->g_h_out, H, wg_t3, H, N, g[FASTEGNN_P_NODE2_W], H, 0, 1, g[FASTEGNN_P_NODE2_B]))) return rc;
  // node_mlp.0: [h | agg | flat(v) | node_attr]
  if ((rc = wb.add(wg_gnp, H, L->h, H, N, g[FASTEGNN_P_NODE0_W], ld_n0, 0, 1, g[FASTEGNN_P_NODE0_B]))) return rc;
  if ((rc = wb.add(wg_gnp, H, L->aggm, H, N, g[FASTEGNN_P_NODE0_W], ld_n0, H, 1, nullptr))) return rc;
  if (!shared && (rc = wb.finish())) return rc;
  if (L->na > 0) {
    if ((rc = launch_wgrad_small(wg_gnp, H, L->node_attr, L->na, L->na, N, g[FASTEGNN_P_NODE0_W], ld_n0, 2 * H + H * C, st))) return rc;
    if (L->g_node_attr)
      if ((rc = launch_dgrad_small(wg_gnp, N, L->na, L->params[FASTEGNN_P_NODE0_W], ld_n0, 2 * H + H * C, L->g_node_attr, 1, st))) return rc;
  }
  return FASTEGNN_OK;
}

}  // namespace fe

#ifdef FE_STAMP
extern "C" int fastegnn_debug_read_vb2_stamps(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::g_vb2_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fe::g_vb2_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int fastegnn_spin_timeouts(int reset) {
  int n = -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(fe::g_vbs_timeouts), sizeof(int)) != hipSuccess) return -1;
  if (reset) {
    const int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fe::g_vbs_timeouts), &z, sizeof(int));
  }
  return n;
}
#ifdef FE_VBS_WATCHDOG
extern "C" int fastegnn_debug_read_vbs_dog(int *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::g_vbs_dog), sizeof(int) * 64) != hipSuccess) return -1;
  if (reset) {
    int z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fe::g_vbs_dog), z, sizeof(z));
  }
  return 0;
}
#endif
