// Backward stages B5..B1 of one E_GCL_vel layer: the hand-derived adjoint of layer_fwd.hip
// (what autograd computes for models/FastEGNN.py:192-223 at utils/train.py:169).
// Math per stage: oracle/factored.py (*_bwd).  Activations are recomputed per tile; the
// operands of the 64x64 weight gradients are written once ([rows,64]) and contracted by the
// generic wgrad_tn kernel (misc.hip).
#include "stages.h"
#ifdef FE_DIAG_NOSTORE
#define WG_STORE(x)
#else
#define WG_STORE(x) x
#endif

namespace fe {

__device__ __forceinline__ Vec vdsilu_mul(const Vec &g, const Vec &z FE_ACT_P) {
  return vmap2(g, z, [=](float a, float b) { return a * dsilu_f(b FE_ACT_A); });
}
__device__ __forceinline__ Vec vmask(const Vec &v, bool keep) { return keep ? v : vzero(); }

// Rank-1 weight gradients are accumulated per lane (D layout) over a wave's tiles.  At kernel end
// they are summed over the 16 items of the tile (shuffles), over the workgroup's waves (LDS) and
// leave the workgroup as one atomic per element: red is a zeroed [n][64] LDS array.
__device__ __forceinline__ void vec_reduce_lds(float *red_row, const Vec &acc, int j, int q) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = jsum(acc.t[t][r]);
      if (j == 0) atomicAdd(&red_row[16 * t + 4 * q + r], s);
    }
}

// =====================================================================================
// B5 graph_post_bwd
// =====================================================================================
struct GraphPostBwdArgs {
  const float *xsum, *HvT, *poolV, *g_Z_out, *g_HvT_out, *wpack, *b5;
  float *g_Z, *g_HvT, *g_poolV, *g_poolX, *wg_u, *wg_gz5, *wg_pm;
  int B, C, flags;
  float act_param = 0.f;
};
__global__ __launch_bounds__(256) void graph_post_bwd_kernel(GraphPostBwdArgs a) {
  const bool bf = a.flags & FASTEGNN_F_BF16;   // bf16 operand mode: the B operand of every product is rounded
  auto rb = [&](const Vec &v) { return bf ? vround(v) : v; };
  const int l = lane_id(), j = l & 15, q = l >> 4;
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  const int M = a.B * a.C, ntiles = (M + 15) >> 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.B * 3 * a.C; i += gridDim.x * blockDim.x) {
    const int b = i / (3 * a.C);
    const float g = a.g_Z_out[i];
    a.g_Z[i] = g;
    a.g_poolX[i] = g / fmaxf(a.xsum[b * 4 + 3], 1.f);
  }
  if (a.flags & FASTEGNN_F_RF) {   // identity on the virtual features: no pooled-message gradient
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M * H; i += gridDim.x * blockDim.x) {
      a.g_HvT[i] = a.g_HvT_out[i];
      a.g_poolV[i] = 0.f;
    }
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
    gemm64(a.wpack + (size_t)I_W5A * IMG, rb(hv), z5);
    gemm64(a.wpack + (size_t)I_W5B * IMG, rb(pm), z5);
    const Vec g_out = vload_row(a.g_HvT_out + (size_t)mc * H, q);
    Vec g_u = vzero();
    gemm64(a.wpack + (size_t)I_W6T * IMG, rb(g_out), g_u);
    const Vec g_z5 = vdsilu_mul(g_u, z5 FE_ACT(a));
    Vec g_hv = (a.flags & FASTEGNN_F_RESIDUAL) ? g_out : vzero();
    gemm64(a.wpack + (size_t)I_W5AT * IMG, rb(g_z5), g_hv);
    Vec g_pm = vzero();
    gemm64(a.wpack + (size_t)I_W5BT * IMG, rb(g_z5), g_pm);
    if (valid) {
      vstore_row(a.wg_u + (size_t)m * H, q, vsilu(z5 FE_ACT(a)));
      vstore_row(a.wg_gz5 + (size_t)m * H, q, g_z5);
      vstore_row(a.wg_pm + (size_t)m * H, q, pm);
      vstore_row(a.g_HvT + (size_t)m * H, q, g_hv);
      vstore_row(a.g_poolV + (size_t)m * H, q, vscale(g_pm, inv));
    }
  }
}

int graph_post_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared) {
  FE_REQUIRE(L->xsum && L->HvT && L->poolV && L->g_Z_out && L->g_HvT_out && L->g_Z && L->g_HvT && L->g_poolV &&
                 L->g_poolX && L->wg_node && L->grads && L->wpack,
             "graph_post_backward: null buffer");
  const long M = (long)L->B * L->C;
  float *wg_gp = L->wg_node + 4 * wg_node_rows(L) * H;   // this stage's operand region of wg_node (kernels.h)
  float *wg_u = wg_gp, *wg_gz5 = wg_gp + M * H, *wg_pm = wg_gp + 2 * M * H;
  GraphPostBwdArgs a{L->xsum, L->HvT, L->poolV, L->g_Z_out, L->g_HvT_out, L->wpack, L->params[FASTEGNN_P_NODEV0_B],
                     L->g_Z, L->g_HvT, L->g_poolV, L->g_poolX, wg_u, wg_gz5, wg_pm, L->B, L->C, L->flags, L->act_param};
  int grid = cdiv(cdiv(M, 16), 4);
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  { ProfScope _ps_graph_post_bwd_kernel(K_GRAPH_POST_BWD, st); hipLaunchKernelGGL(graph_post_bwd_kernel, dim3(grid), dim3(256), 0, st, a); }
  int rc = check_launch("graph_post_bwd_kernel");
  if (rc) return rc;
  if (has(L, FASTEGNN_F_RF)) return FASTEGNN_OK;   // no node_mlp_virtual
  float *const *g = L->grads;
  WgradBatch local(L->wg_slab, st);
  WgradBatch &wb = shared ? *shared : local;
  wb.round = has(L, FASTEGNN_F_BF16);
  // node_mlp_virtual.2: dW6 += g_out^T u, db6 += colsum g_out
  if ((rc = wb.add(L->g_HvT_out, H, wg_u, H, M, g[FASTEGNN_P_NODEV2_W], H, 0, 1, g[FASTEGNN_P_NODEV2_B]))) return rc;
  // node_mlp_virtual.0: [Hv | pooled v]
  if ((rc = wb.add(wg_gz5, H, L->HvT, H, M, g[FASTEGNN_P_NODEV0_W], 2 * H, 0, 1, g[FASTEGNN_P_NODEV0_B]))) return rc;
  if ((rc = wb.add(wg_gz5, H, wg_pm, H, M, g[FASTEGNN_P_NODEV0_W], 2 * H, H, 1, nullptr))) return rc;
  return shared ? FASTEGNN_OK : wb.finish();
}

// B4 (virt_backward: the adjoint of the virtual stage and of node_model) lives in virt_bwd.hip

// =====================================================================================
// B3 graph_pre_bwd: adjoint of Bc / Gram / centroid
// =====================================================================================
struct GraphPreBwdArgs {
  const float *xsum, *Z, *g_Bc, *g_Zp, *V0W;
  float *g_Z, *g_HvT, *g_xbar, *wg_mxt;
  int B, C, bf16;
};
// grid (B, GPB_SPLIT): the per-(channel, feature) outputs of a graph are dealt to GPB_SPLIT workgroups; each of them
// recomputes the graph's small matrices (mz, g_mX, g_mz), workgroup 0 writes the centroid gradient
constexpr int GPB_SPLIT = 4;
__global__ __launch_bounds__(256) void graph_pre_bwd_kernel(GraphPreBwdArgs a) {
  extern __shared__ float sm[];
  const int C = a.C, b = blockIdx.x, ld = 2 * H + 1 + C;
  const int i0 = blockIdx.y * 256 + threadIdx.x, istep = 256 * gridDim.y;
  float *mz = sm;               // [3][C]
  float *gmX = sm + 3 * C;      // [C][C]  d/d mX[c'][c] stored at [c'*C + c]
  float *gmz = gmX + C * C;     // [3][C]
  const float cnt = fmaxf(a.xsum[b * 4 + 3], 1.f);
  for (int i = threadIdx.x; i < 3 * C; i += 256) mz[i] = a.Z[(size_t)b * 3 * C + i] - a.xsum[b * 4 + i / C] / cnt;
  __syncthreads();
  // mX^T rows (feature vector of channel c = column c of mX), zero padded to 64, for dV1d
  for (int i = i0; i < C * H; i += istep) {
    int c = i >> 6, d = i & 63;
    float v = 0.f;
    if (d < C) v = mz[c] * mz[d] + mz[C + c] * mz[C + d] + mz[2 * C + c] * mz[2 * C + d];
    a.wg_mxt[((size_t)b * C + c) * H + d] = v;
  }
  // g_HvT[b,c,k] += sum_o g_Bc[b,c,o] V1b[o,k]
  for (int i = i0; i < C * H; i += istep) {
    int c = i >> 6, k = i & 63;
    const float *gb = a.g_Bc + ((size_t)b * C + c) * H;
    float acc = 0.f;
    if (a.bf16) { for (int o = 0; o < H; ++o) acc += round_bf(gb[o]) * round_bf(a.V0W[(size_t)o * ld + H + k]); }   // V1b^T g_Bc: bf16 operands
    else { for (int o = 0; o < H; ++o) acc += gb[o] * a.V0W[(size_t)o * ld + H + k]; }
    a.g_HvT[((size_t)b * C + c) * H + k] += acc;
  }
  // g_mX[c'][c] = sum_o g_Bc[b,c,o] V1d[o,c']   (every workgroup: all of it)
  for (int i = threadIdx.x; i < C * C; i += 256) {
    int cp = i / C, c = i % C;
    const float *gb = a.g_Bc + ((size_t)b * C + c) * H;
    float acc = 0.f;
    for (int o = 0; o < H; ++o) acc += gb[o] * a.V0W[(size_t)o * ld + 2 * H + 1 + cp];
    gmX[cp * C + c] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    int k = i / C, c = i % C;
    float acc = 0.f;
    for (int d = 0; d < C; ++d) acc += (gmX[c * C + d] + gmX[d * C + c]) * mz[k * C + d];
    gmz[i] = acc;
    if (i % gridDim.y == blockIdx.y) a.g_Z[(size_t)b * 3 * C + i] += acc + a.g_Zp[(size_t)b * 3 * C + i];
  }
  __syncthreads();
  if (threadIdx.x < 3 && blockIdx.y == 0) {
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += gmz[threadIdx.x * C + c];
    a.g_xbar[b * 3 + threadIdx.x] = -acc / cnt;
  }
}
int graph_pre_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared) {
  FE_REQUIRE(L->xsum && L->Z && L->HvT && L->g_Bc && L->g_Zp && L->g_Z && L->g_HvT && L->g_xbar && L->wg_node && L->grads,
             "graph_pre_backward: null buffer");
  const long M = (long)L->B * L->C;
  float *wg_mxt = L->wg_node + 7 * wg_node_rows(L) * H;   // this stage's operand region of wg_node (kernels.h)
  GraphPreBwdArgs a{L->xsum, L->Z, L->g_Bc, L->g_Zp, L->params[FASTEGNN_P_VIRT0_W], L->g_Z, L->g_HvT, L->g_xbar, wg_mxt,
                    L->B, L->C, has(L, FASTEGNN_F_BF16) ? 1 : 0};
  const size_t lds = (size_t)(6 * L->C + L->C * L->C) * sizeof(float);
  { ProfScope _ps_graph_pre_bwd_kernel(K_GRAPH_PRE_BWD, st); hipLaunchKernelGGL(graph_pre_bwd_kernel, dim3(L->B, GPB_SPLIT), dim3(256), lds, st, a); }
  int rc = check_launch("graph_pre_bwd_kernel");
  if (rc) return rc;
  float *const *g = L->grads;
  const int ld = 2 * H + 1 + L->C;
  // edge_mlp_virtual.0: columns [H,2H) <- Hv, columns [2H+1, 2H+1+C) <- mX[:,c], bias
  WgradBatch local(L->wg_slab, st);
  WgradBatch &wb = shared ? *shared : local;
  wb.round = has(L, FASTEGNN_F_BF16);
  if ((rc = wb.add(L->g_Bc, H, L->HvT, H, M, g[FASTEGNN_P_VIRT0_W], ld, H, 1, g[FASTEGNN_P_VIRT0_B]))) return rc;
  wb.round = false;   // the Gram columns of edge_mlp_virtual.0 are an fp32 product in every mode
  if ((rc = wb.add(L->g_Bc, H, wg_mxt, H, M, g[FASTEGNN_P_VIRT0_W], ld, 2 * H + 1, 1, nullptr, 1, 0, 0, 0, L->C))) return rc;
  return shared ? FASTEGNN_OK : wb.finish();
}

// =====================================================================================
// B2 edge_bwd
// =====================================================================================
struct EdgeBwdArgs {
  EdgeArgs f;
  const float *g_aggm, *g_aggx;
  float *g_P, *g_xrow, *g_QXe;
  float *g_QXs_atomic;   // non-null: scatter d/d(Q|x) straight into the source table with float atomics (no g_QXe, no CSC reduce)
  float *g_ea;   // [E,ea] d loss / d edge_attr (sorted-edge order, +=) or null
  float *d_wx2, *d_attw, *d_attb, *d_bx2;
  float *d_wr, *d_we;   // edge_mlp.0.weight grad: radial column and first edge_attr column (row stride ld_e0)
  int ld_e0, C;
  float *slab, *slab_b;   // producer/consumer variant: partial slabs of the two in-kernel weight gradients
  int slab_w2, slab_wx1;
  float *cons_scratch;    // [grid][2][64*64] running sums of the two consumers, accumulator order (wg_edge)
};
#ifndef FE_PC_PRIO
#define FE_PC_PRIO 3
#endif

// ---- producer/consumer variant: the two 64x64 weight gradients of the edge stage are contracted inside the
// workgroup.  Six waves run the tile adjoint (producers) and hand each operand pair (g_mp,t) / (g_up,m) as two
// 16x64 tiles to the consumer wave of that weight through a ring of LDS slots; a consumer owns one 64x64 accumulator and
// writes one partial slab per workgroup and weight (summed by wgrad_reduce_kernel in a fixed order).  The
// operands never reach HBM.  The consumer contracts two slots of a ring at a time (K = 32 edges) as bf16x3
// products on the matrix pipe.  Slot protocol (tickets taken from an LDS counter per ring, consumed in order):
// producer waits drained[s] == round, writes, sets filled[s] = round + 1; consumer waits filled[s] == round + 1,
// contracts, sets drained[s] = round + 1.
// Eight waves: waves are dealt round-robin to the four SIMDs, so waves 3 and 7 share SIMD 3 and have it to themselves --
// they are the two CONSUMERS, one per ring (wave 3: edge_mlp.2, wave 7: coord_mlp_r.0); the six producers sit two per SIMD
// on SIMDs 0..2.  A single consumer serving both rings was latency-exposed (dependent LDS reads -> split -> MFMA chain at
// one wave on its SIMD): with the contractions skipped the kernel ran 18 % faster (-DFE_DIAG_NOCONS).
// -DFE_PC_WAVES=12 (round 6 experiment): three waves per SIMD -- ten producers (waves 0,1,2,4,5,6,8,9,10,11) + the two consumers, 168 registers
#ifndef FE_PC_WAVES
#define FE_PC_WAVES 8
#endif
constexpr int PC_WAVES = FE_PC_WAVES;
constexpr int PC_CONS = 3, PC_CONS2 = 7;
constexpr int PC_PROD = PC_WAVES - 2;       // producer waves
// Round 5, f16x2 build: the consumers contract on f16x2 products with a sticky scale and 32x32x16 MFMAs (common.h, WgAcc32), one
// ticket (16 edges) per step.  -DFE_PC_CONS32=0 restores the bf16x3 consumers.
#ifndef FE_PC_CONS32
#define FE_PC_CONS32 1
#endif
// (round 4, with the f16x2 producers, two repeats on one box, tools/gpu_ab_rings2.sh: 2 slots 3.23-3.26, 3 slots 3.16-3.20,
//  4 slots 3.25-3.27, 5 slots 3.24 ms per step; round 2 had gone from 2 to 4 with the bf16x3 producers)
#ifndef FE_PC_RING
#define FE_PC_RING 3
#endif
constexpr int PC_RING = FE_PC_RING;                  // slots per ring; one ring per weight (kind 0: edge_mlp.2, kind 1: coord_mlp_r.0)
#ifndef FE_PC_FLUSH
#define FE_PC_FLUSH 48
#endif
constexpr int PC_FLUSH = FE_PC_FLUSH;       // tickets (16-edge operand sets) a consumer accumulates in registers between two scratch updates
constexpr int PC_RS = 68;                   // row stride of a slot tile
constexpr int PC_SLOT = 2 * 16 * PC_RS;     // floats per slot: G tile | T tile
// W2 | WX1 as row-major split images, each serving the product and its transpose (an f16x2 image keeps two parts in LDS)
template <int MODE> constexpr int pc_img_floats() { return 2 * (rm_lds_bytes<MODE>() / 4); }
enum { PC_HEAD = 0, PC_TOTAL = 2, PC_FILLED = 4, PC_DRAINED = 4 + 2 * PC_RING, PC_CTRL = 4 + 4 * PC_RING };
#ifdef FE_SAFE_WAITS   // ring flags as workgroup-scope acquire loads / release stores instead of relaxed accesses between fences
__device__ __forceinline__ int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
__device__ __forceinline__ int lds_ld(const int *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void lds_st(int *p, int v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
#endif
__device__ __forceinline__ void pc_tile_store(float *tile, int j, int q, const Vec &v) {
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(tile + j * PC_RS + 16 * t + 4 * q) = v.t[t];
}
// EA: edge_attr slots whose weight-column sums are accumulated in the row walk, without guards (2 covers edge_attr_nf <= 2 --
// every BASELINE configuration --, 7 the rest): a per-slot `k < ea_dim` test inside the 16-edge walk compiled into ~130
// scalar branches per tile and made the walk 23 % of the producers' time (phase stamps).
template <int MODE, int EA>
__global__ __launch_bounds__(64 * PC_WAVES) void edge_bwd_pc_kernel(EdgeBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef FE_ISA_CONST   // assembly-only builds of tools/isa_budget_bwd.py: the layer flags as a constant (straight-line code of ONE configuration)
  A.f.flags = FE_ISA_CONST; A.f.ea_dim = 2; A.f.bx2 = nullptr; A.f.attw = nullptr; A.g_ea = nullptr; A.d_bx2 = nullptr;
#endif
  const EdgeArgs &a = A.f;
  float *img = lds;                    // W2, WX1 (row-major split images, common.h)
  float *vec = lds + pc_img_floats<MODE>();
  float *tiles = vec + EV_COUNT * H;   // per producer [16][TS]; the 4 pad columns of a row hold its g_d scalars
  float *ring = tiles + PC_PROD * 16 * TS;
  int *ctrl = reinterpret_cast<int *>(ring + 2 * PC_RING * PC_SLOT);
  {
    // slots 3, 4 (f16x2: 10, 11) are consecutive in wpack, RM_BYTES apart; only the parts this form reads are copied
    constexpr int RMS = rm_lds_bytes<MODE>();
    const char *src = wpack_rm(a.wpack, A.C, (MODE == GM_F16 ? RM_F16 : 0) + 3);
    for (int i = threadIdx.x; i < 2 * (RMS / 16); i += blockDim.x) {
      const int im = i / (RMS / 16), k = i % (RMS / 16);
      reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(img) + im * RMS)[k] = reinterpret_cast<const u32x4 *>(src + (size_t)im * RM_BYTES)[k];
    }
  }
  edge_load_vecs(vec, a);
  if (threadIdx.x < PC_CTRL) ctrl[threadIdx.x] = 0;
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4, wv = wave_id();
  const bool consumer = wv == PC_CONS || wv == PC_CONS2;
  const int ckind = wv == PC_CONS ? 0 : 1;     // the ring a consumer wave serves
  const int pw = wv - (wv > PC_CONS ? 1 : 0) - (wv > PC_CONS2 ? 1 : 0);   // producer index (waves 0,1,2,4,5,6[,8..11] -> 0..5[..9])
  float *pt = tiles + (consumer ? 0 : pw) * 16 * TS;
  const int wave = (int)blockIdx.x * PC_PROD + pw, nwaves = (int)gridDim.x * PC_PROD;
  const bool mean = !(a.flags & FASTEGNN_F_COORDS_SUM);
  float accW[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // [radial | edge_attr] columns of edge_mlp.0, lane = out
  const bool tanh_on = a.flags & FASTEGNN_F_TANH, att_on = a.flags & FASTEGNN_F_ATTENTION,
             norm_on = a.flags & FASTEGNN_F_NORMALIZE;
  Vec acc_wx2 = vzero(), acc_att = vzero();
  float acc_attb = 0.f, acc_bx2 = 0.f;
  FE_T0()
  // this wave's share: a contiguous run of whole rows holding ~E/nwaves edges (see edge_fwd_kernel)
  int r0 = 0, r1 = 0, e0 = 0, e1 = 0;
  if (!consumer) {
    const int c0 = (int)((long)wave * a.n_chunks / nwaves), c1 = (int)((long)(wave + 1) * a.n_chunks / nwaves);
    r0 = a.chunk_row[c0];
    r1 = a.chunk_row[c1];
    if (r0 < r1) {
      e0 = a.rowptr[r0];
      e1 = a.rowptr[r1];
      if (l == 0) atomicAdd(&ctrl[PC_TOTAL], (e1 - e0 + 15) >> 4);   // tiles = tickets per ring
    }
  }
  __syncthreads();
  // hand one operand pair to the consumer
  auto publish = [&](int kind, const Vec &Gv, const Vec &Tv) {
    int tk = 0;
    if (l == 0) tk = atomicAdd(&ctrl[PC_HEAD + kind], 1);
    tk = __builtin_amdgcn_readfirstlane(tk);
    const int sl = tk % PC_RING, round = tk / PC_RING;
    while (lds_ld(&ctrl[PC_DRAINED + kind * PC_RING + sl]) != round) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // (compiler order: the tile stores stay behind the flag read)
    float *slot = ring + (kind * PC_RING + sl) * PC_SLOT;
    pc_tile_store(slot, j, q, Gv);
    pc_tile_store(slot + 16 * PC_RS, j, q, Tv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // lgkmcnt(0): the tile is in LDS before the flag
    if (l == 0) lds_st(&ctrl[PC_FILLED + kind * PC_RING + sl], round + 1);
  };
  constexpr bool CONS32 = MODE == GM_F16 && FE_PC_CONS32 != 0;
  if (CONS32 && consumer) {
    if constexpr (CONS32) {
#if FE_PC_PRIO
    __builtin_amdgcn_s_setprio(FE_PC_PRIO);
#endif
    // f16x2 consumer (common.h: WgAcc32 / WgScale): ring `ckind`, one ticket per step, the slot handed back as soon as its values are
    // in registers; sticky power-of-two scales per operand stream; running sums in TRUE units in this wave's scratch tile
    const int total = lds_ld(&ctrl[PC_TOTAL]);
    WgAcc32 acc;
    wg32_zero(acc);
    WgScale sG{0}, sT{0};
    double bs[2] = {0., 0.};
    auto scp = [&](int blk, int e4) {
      char *b = reinterpret_cast<char *>(A.cons_scratch + ((size_t)blockIdx.x * 2 + ckind) * IMG) + (size_t)((blk * 4 + e4) * 64 * 16);
      asm volatile("" : "+s"(b));
      return reinterpret_cast<f32x4 *>(b + (unsigned)l * 16u);
    };
    bool flushed = false;
    const size_t sl = (size_t)(ckind == 0 ? A.slab_w2 : A.slab_wx1) + blockIdx.x;
    auto flush = [&](bool last) {
      const float ig = sG.inv(), it = sT.inv();
      float *slab_dst = A.slab + sl * IMG;
#pragma unroll
      for (int bo = 0; bo < 2; ++bo)
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (acc.c[bo][bk][4 * e4 + r] * ig) * it;
            f32x4 *d = scp(bo * 2 + bk, e4);
            if (flushed) v += *d;
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
    int since = 0;
    for (int done = 0; done < total; ++done) {
      const int s0 = done % PC_RING, r0w = done / PC_RING;
      while (lds_ld(&ctrl[PC_FILLED + ckind * PC_RING + s0]) != r0w + 1) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      const float *g0 = ring + (ckind * PC_RING + s0) * PC_SLOT;
      float xg[2][8], xt[2][8];
      wg32_read<PC_RS>(g0, xg);
      wg32_read<PC_RS>(g0 + 16 * PC_RS, xt);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (l == 0) lds_st(&ctrl[PC_DRAINED + ckind * PC_RING + s0], r0w + 1);
      const float f = sG.update_lazy(xg) * sT.update_lazy(xt);
      if (f != 1.f) wg32_scale_acc(acc, f);
      const WgOp32 G = wg32_split(xg, sG.scale()), T = wg32_split(xt, sT.scale());
#pragma unroll
      for (int b = 0; b < 2; ++b)
        bs[b] += (double)(((xg[b][0] + xg[b][1]) + (xg[b][2] + xg[b][3])) + ((xg[b][4] + xg[b][5]) + (xg[b][6] + xg[b][7])));
      wg32_mma(acc, G, T);
      if (++since >= PC_FLUSH && done + 1 < total) {
        flush(false);
        flushed = true;
        since = 0;
      }
    }
    flush(true);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      double s0 = bs[b];
      s0 += __shfl_xor(s0, 32);
      if (l < 32) A.slab_b[sl * H + 32 * b + l] = (float)s0;
    }
    }
  } else if (consumer) {
#if FE_PC_PRIO
    __builtin_amdgcn_s_setprio(FE_PC_PRIO);   // the consumer must never be the slower side: it wins issue arbitration on its SIMD
#endif
    const int total = lds_ld(&ctrl[PC_TOTAL]);   // tiles of this workgroup = tickets per ring
    f32x4 accA[4][4];
    double bsA[4] = {0., 0., 0., 0.};   // bias column sums: eight rows at a time in fp32, everything across tickets in double
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tk = 0; tk < 4; ++tk) accA[ti][tk] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K = 32 edges per step: tickets tk, tk+1 of one ring; lane (q,i) takes feature i of the rows 4q+e of the first
    // (e < 4) and of the second slot (e >= 4), the same map for both operands.  With the 68-float row stride the
    // rows of the two quarter-waves read together sit 16 banks apart: conflict-free b32 reads
    auto contract = [&](int kind, int tk, f32x4 (&acc)[4][4], double (&bs)[4]) {
      const int s0 = tk % PC_RING, r0w = tk / PC_RING;
      const bool two = tk + 1 < total;
      const int s1 = (tk + 1) % PC_RING, r1w = (tk + 1) / PC_RING;
      while (lds_ld(&ctrl[PC_FILLED + kind * PC_RING + s0]) != r0w + 1) __builtin_amdgcn_s_sleep(2);
      if (two)
        while (lds_ld(&ctrl[PC_FILLED + kind * PC_RING + s1]) != r1w + 1) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // the slot reads stay behind the flag reads
      const float *g0 = ring + (kind * PC_RING + s0) * PC_SLOT, *g1 = ring + (kind * PC_RING + s1) * PC_SLOT;
#ifdef FE_DIAG_NOCONS   // diagnostic: the consumer only drains its slots (is the kernel consumer-bound?)
      if (l == 0) {
        lds_st(&ctrl[PC_DRAINED + kind * PC_RING + s0], r0w + 1);
        if (two) lds_st(&ctrl[PC_DRAINED + kind * PC_RING + s1], r1w + 1);
      }
      (void)g0; (void)g1; (void)acc; (void)bs;
      return;
#endif
      // every value of the two slots is read first and the slots are handed back BEFORE the splits and products: with
      // two slots per ring the producers otherwise wait out the whole contraction (their `publish` phases were 17 % of
      // the producer time in the stamps)
      float xb[4][8], xa[4][8];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xb[t][e] = g0[16 * PC_RS + (4 * q + e) * PC_RS + 16 * t + j];
          xb[t][4 + e] = two ? g1[16 * PC_RS + (4 * q + e) * PC_RS + 16 * t + j] : 0.f;
          xa[t][e] = g0[(4 * q + e) * PC_RS + 16 * t + j];
          xa[t][4 + e] = two ? g1[(4 * q + e) * PC_RS + 16 * t + j] : 0.f;
        }
      // every read of the slots has returned (lgkmcnt(0)) and, for the compiler, none of them may sink below the hand-back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (l == 0) {
        lds_st(&ctrl[PC_DRAINED + kind * PC_RING + s0], r0w + 1);
        if (two) lds_st(&ctrl[PC_DRAINED + kind * PC_RING + s1], r1w + 1);
      }
      Split8 B[4];   // bf16 mode: only .h is used (RNE-rounded operand, one product)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (MODE == GM_BF16) B[t].h = round8(xb[t]);
        else B[t] = split8(xb[t]);
      }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const float (&x)[8] = xa[ti];
        bs[ti] += (double)(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7])));
        if constexpr (MODE == GM_BF16) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, round8(x));
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[ti][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, B[t].h), acc[ti][t], 0, 0, 0);
          continue;
        }
        const Split8 Aop = split8(x);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, Aop.h), am = __builtin_bit_cast(bf16x8, Aop.m),
                     al = __builtin_bit_cast(bf16x8, Aop.l);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, B[t].h), bm = __builtin_bit_cast(bf16x8, B[t].m),
                       bl = __builtin_bit_cast(bf16x8, B[t].l);
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
    };
    // this wave's ring: its next K-step is complete with two filled slots, or with the last single one
    auto ready = [&](int kind, int done) {
      const int s0 = done % PC_RING, s1 = (done + 1) % PC_RING;
      if (lds_ld(&ctrl[PC_FILLED + kind * PC_RING + s0]) != done / PC_RING + 1) return false;
      return done + 1 >= total || lds_ld(&ctrl[PC_FILLED + kind * PC_RING + s1]) == (done + 1) / PC_RING + 1;
    };
    // The accumulator leaves the registers every PC_FLUSH tickets (768 edges): ONE fp32 chain over all of a workgroup's edges
    // (7 500 at cfg4) carries 2-4x the rounding noise of 780-row chains on cancelling sums (measured on the virtual stage,
    // virt_bwd.hip).  The running sum lives in a scratch tile of this wave in accumulator order (sixteen 16-byte
    // read-modify-writes per lane, L2 resident); the [o][k] slab is written once, at the end.
    f32x4 *sc = reinterpret_cast<f32x4 *>(A.cons_scratch + ((size_t)blockIdx.x * 2 + ckind) * IMG) + l;
    double bs_tot[4] = {0., 0., 0., 0.};   // bias column sums: fp32 between two flushes, double across them (virt_bwd.hip)
    bool flushed = false;
    int done = 0, since = 0;
    while (done < total) {
      if (ready(ckind, done)) {
        contract(ckind, done, accA, bsA);
        done += 2;
        since += 2;
        if (since >= PC_FLUSH && done < total) {
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
            for (int tk2 = 0; tk2 < 4; ++tk2) {
              f32x4 *d = sc + (ti * 4 + tk2) * 64;
              if (flushed) accA[ti][tk2] += *d;
              *d = accA[ti][tk2];
              accA[ti][tk2] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            bs_tot[ti] += bsA[ti];
            bsA[ti] = 0.;
          }
          flushed = true;
          since = 0;
        }
      } else {
        __builtin_amdgcn_s_sleep(1);
      }
    }
    // one partial slab per workgroup and weight: [o][k] row-major, o = G feature, k = T feature
    const size_t sl = (size_t)(ckind == 0 ? A.slab_w2 : A.slab_wx1) + blockIdx.x;
    float *sa = A.slab + sl * IMG;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tk2 = 0; tk2 < 4; ++tk2) {
        f32x4 v = accA[ti][tk2];
        if (flushed) v += sc[(ti * 4 + tk2) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) sa[(16 * ti + 4 * q + r) * H + 16 * tk2 + j] = v[r];
      }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      double s0 = bs_tot[ti] + bsA[ti];
      s0 += __shfl_xor(s0, 16);
      s0 += __shfl_xor(s0, 32);
      if (q == 0) A.slab_b[sl * H + 16 * ti + j] = (float)s0;
    }
  }
  if (!consumer && r0 < r1) {
    int cur = -1;
    float acc = 0.f, accx = 0.f;
    auto flush = [&]() {
      A.g_P[(size_t)cur * H + l] = acc;
      if (l < 3) A.g_xrow[(size_t)cur * 3 + l] = accx;   // lanes 0..2 hold x,y,z (lane 3: pad)
    };
    // rows without edges inside this wave's range [r0, r1) are zeroed here (no memset ahead of the kernel)
    auto zero_rows = [&](int ra, int rb) {
      for (int r = ra; r < rb; ++r) {
        A.g_P[(size_t)r * H + l] = 0.f;
        if (l < 3) A.g_xrow[(size_t)r * 3 + l] = 0.f;
      }
    };
    // the indices of a tile are requested one tile ahead (4 registers at EA = 2): of the two dependent round trips at the
    // head of a tile (index -> gathered rows / coordinates; 4.1 k + 1.8 k of 26 k cycles in the stamps) the first is gone
    // (the generic edge_attr variant, EA = 7, would need 9 registers and spill: it keeps the load at the head of the tile)
    constexpr bool PF_IDX = EA <= 2;
    EdgeIdx nxt_i;
    if (PF_IDX && e0 < e1) edge_load_idx(a, min(e0 + j, e1 - 1), nxt_i);
    // (requesting the gathered rows ahead as well -- before the row walk that ends a tile -- costs 40 live registers
    // there and spills 24-36 of them: not done)
    for (int base = e0; base < e1; base += 16) {
      asm volatile("" ::: "memory");
      const int nvalid = min(16, e1 - base);
      const bool valid = j < nvalid;
      const int e = min(base + j, e1 - 1);
      EdgeIdx cur_i;
      if constexpr (PF_IDX) {
        cur_i = nxt_i;
        if (base + 16 < e1) edge_load_idx(a, min(base + 16 + j, e1 - 1), nxt_i);
      } else {
        edge_load_idx(a, e, cur_i);
      }
      EdgeFwdState S;
      Vec pre;
      // the row-keyed operands of the adjoint (degree, g_aggx row, g_aggm row) depend on the row index only: requested here,
      // they arrive under the forward recompute instead of being waited for one after the other in the middle of the tile
      const int rp0 = a.rowptr[cur_i.row], rp1 = a.rowptr[cur_i.row + 1];
      float gax[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) gax[k] = A.g_aggx[(size_t)cur_i.row * 3 + k];
      Vec gam;   // (EA = 7 variant: 16 more live registers would spill; it keeps this row's load at its use)
      constexpr bool GAM_EARLY = PF_IDX && PC_WAVES <= 8;   // (12 waves: 168 registers -- the row is requested at its use)
      if constexpr (GAM_EARLY) gam = vload_row(A.g_aggm + (size_t)cur_i.row * H, q);
      edge_tile_forward<true, MODE, true>(a, img, vec, cur_i, q, S, pre FE_TA);   // pre, S.mp, S.up now hold silu'()
      const int dg = rp1 - rp0;
      const float inv = valid ? rcp_f((float)(dg > 1 ? dg : 1)) : 0.f;
      const float invx = valid ? (mean ? inv : 1.f) : 0.f;
      tile_store(pt, j, q, S.t);   // parked until its partner g_mp exists (the tile is free until the row walk)
      // coordinate head adjoint (coord_mlp_r, :125)
      float g_tr[3], g_dn[3], g_s = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        g_tr[k] = gax[k] * invx;
        g_s += S.dn[k] * g_tr[k];
        g_dn[k] = S.s * g_tr[k];
      }
      const float g_sr = tanh_on ? g_s * (1.f - S.s * S.s) : g_s;
      vaxpy(acc_wx2, g_sr, S.u);
      if (q == 0) acc_bx2 += g_sr;
      const Vec g_up = vmul(vscale(vload_vec(vec + EV_WX2 * H, q), g_sr), S.up);
      FE_T(5)   // degree / g_aggx rows, head adjoint, g_up
      publish(1, g_up, S.m);
      FE_T(6)   // publish (g_up, m)
      if constexpr (!GAM_EARLY) gam = vload_row(A.g_aggm + (size_t)S.row * H, q);
      Vec g_m = vscale(gam, inv);
      gemm_e<MODE, 3, true>(img, g_up, g_m);
      Vec g_m0 = g_m;
      if (att_on) {
        const float g_a = vdot(g_m, S.m0);
        const float g_z = g_a * S.att * (1.f - S.att);
        vaxpy(acc_att, g_z, S.m0);
        if (q == 0) acc_attb += g_z;
        g_m0 = vscale(g_m, S.att);
        vaxpy(g_m0, g_z, vload_vec(vec + EV_ATT * H, q));
      }
      const Vec g_mp = vmul(g_m0, S.mp);
      {
        Vec tpark;
#pragma unroll
        for (int t = 0; t < 4; ++t) tpark.t[t] = *reinterpret_cast<const f32x4 *>(pt + j * TS + 16 * t + 4 * q);
        FE_T(7)   // g_aggm row, WX1^T product, attention adjoint, g_mp
        publish(0, g_mp, tpark);
        FE_T(8)   // publish (g_mp, t)
      }
      Vec g_t = vzero();
      gemm_e<MODE, 2, true>(img, g_mp, g_t);
      const Vec g_pre = vmul(g_t, pre);
      float g_r = vdot(g_pre, vload_vec(vec + EV_WR * H, q));
      if (a.flags & FASTEGNN_F_EGNN_NORM) g_r *= S.r >= 1e-12f ? 0.f : 1e12f;   // d normalize(r^2) / d r^2
      float g_d[3];
      const float invn = norm_on ? rcp_f(S.nrm + a.eps) : 1.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) g_d[k] = g_dn[k] * invn + 2.f * g_r * S.d[k];
      if (valid && !A.g_QXs_atomic) {
        float *qe = A.g_QXe + (size_t)e * QXLD;
        vstore_row(qe, q, g_pre);
        if (q == 0) *reinterpret_cast<f32x4 *>(qe + H) = f32x4{-g_d[0], -g_d[1], -g_d[2], 0.f};
      }
      if (A.g_ea) {   // d/d edge_attr[e,k] = <g_pre[e], edge_mlp.0.weight[:, 2H+1+k]>  (wave-uniform: only when asked for)
        for (int k = 0; k < a.ea_dim; ++k) {
          const float ge = vdot(g_pre, vload_vec(vec + (EV_WE + k) * H, q));
          if (valid && q == 0) A.g_ea[(size_t)e * a.ea_dim + k] += ge;
        }
      }
      FE_T(9)   // W2^T product, g_pre, g_d, per-edge stores
      // row-side segment sums: g_P[row] = sum g_pre, g_xrow[row] = sum g_d
      tile_store(pt, j, q, g_pre);
      if (q == 0) *reinterpret_cast<f32x4 *>(pt + j * TS + H) = f32x4{g_d[0], g_d[1], g_d[2], 0.f};
      __builtin_amdgcn_wave_barrier();
      float mv[16], xv[16];
#pragma unroll
      for (int ee = 0; ee < 16; ++ee) {   // all LDS reads up front; the walk below runs on registers
        mv[ee] = pt[ee * TS + l];
        xv[ee] = pt[ee * TS + H + (l & 3)];
      }
      const int rowv = S.row;
#ifdef FE_DIAG_NOATOMIC   // diagnostic (wrong col-side gradients): what do the scatter atomics cost the producers?  (round 6: 14 % of the kernel)
      if (false) {
#else
      if (A.g_QXs_atomic) {   // default (no FASTEGNN_F_DETERMINISTIC): one coalesced 256-byte atomic row per edge ...
#endif
        const int colv = S.col;
        char *gb = reinterpret_cast<char *>(A.g_QXs_atomic);   // wave-uniform base + 32-bit lane offsets (tables < 2^30 floats)
        const unsigned lo = 4u * (unsigned)l;
        if (nvalid == 16) {
#pragma unroll
          for (int ee = 0; ee < 16; ++ee)
            atomicAdd(reinterpret_cast<float *>(gb + ((unsigned)__builtin_amdgcn_readlane(colv, ee) * (QXLD * 4u) + lo)), mv[ee]);
        } else {
          for (int ee = 0; ee < nvalid; ++ee)
            atomicAdd(reinterpret_cast<float *>(gb + ((unsigned)__shfl(colv, ee) * (QXLD * 4u) + lo)), pt[ee * TS + l]);
        }
        // ... and ONE atomic for the tile's coordinate parts: lane l < 3 * nvalid adds component l % 3 of edge l / 3
        const int ex = min(l / 3, 15), kx = l - 3 * ex;
        const unsigned cx = (unsigned)__shfl(colv, ex);   // lane ex (q = 0) holds edge ex
        const float gx = pt[ex * TS + H + (kx & 3)];
        if (l < 3 * nvalid) atomicAdd(reinterpret_cast<float *>(gb + (cx * (QXLD * 4u) + 4u * (unsigned)(H + kx))), -gx);
      }
      // row changes inside the tile from one DPP compare + ballot (as in edge_fwd_kernel); the row id is read only at a change
#ifndef FE_WALK_READLANE
      const int prevrow = __builtin_amdgcn_update_dpp(rowv, rowv, 0x111, 0xf, 0xf, false);   // row_shr:1
      const unsigned starts = (unsigned)__builtin_amdgcn_ballot_w64(j == 0 ? rowv != cur : rowv != prevrow) & 0xffffu;
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
          }
          const float gp = mv[ee];
          acc += gp;
          accx += xv[ee];
          // d edge_mlp.0.weight[:, 2H + k] += g_pre * [radial | edge_attr][k]   (lane = output row);
          // the per-edge scalars come from the owning lane's registers (v_readlane), not from LDS
          accW[0] += gp * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S.rf), ee));
#pragma unroll
          for (int k = 0; k < EA; ++k)   // slots beyond ea_dim hold zeros
            accW[1 + k] += gp * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S.eav[k]), ee));
        }
      }
      __builtin_amdgcn_wave_barrier();
      FE_T(10)   // transpose tile + row-segmented sums
    }
    if (cur >= 0) flush();
    zero_rows(cur >= 0 ? cur + 1 : r0, r1);
    FE_TEND()
  }
  float *red = vec;   // [3 + 8][64]; the weight vectors are dead now
  __syncthreads();
  for (int i = threadIdx.x; i < 11 * H; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k <= a.ea_dim) atomicAdd(&red[(3 + k) * H + l], accW[k]);
  vec_reduce_lds(red, acc_wx2, j, q);
  if (att_on) {
    vec_reduce_lds(red + H, acc_att, j, q);
    float s = jsum(acc_attb);
    if (l == 0) atomicAdd(&red[2 * H], s);
  }
  if (A.d_bx2) {
    float s = jsum(acc_bx2);
    if (l == 0) atomicAdd(&red[2 * H + 1], s);
  }
  __syncthreads();
  if (threadIdx.x < H) {
    atomicAdd(&A.d_wx2[threadIdx.x], red[threadIdx.x]);
    if (att_on) {
      atomicAdd(&A.d_attw[threadIdx.x], red[H + threadIdx.x]);
      if (threadIdx.x == 0) atomicAdd(A.d_attb, red[2 * H]);
    }
    if (A.d_bx2 && threadIdx.x == 0) atomicAdd(A.d_bx2, red[2 * H + 1]);
    atomicAdd(&A.d_wr[(size_t)threadIdx.x * A.ld_e0], red[3 * H + threadIdx.x]);
    for (int k = 0; k < a.ea_dim; ++k)
      atomicAdd(&A.d_we[(size_t)threadIdx.x * A.ld_e0 + k], red[(4 + k) * H + threadIdx.x]);
  }
}

int edge_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared) {
  FE_REQUIRE(L->P && L->QX && L->g_aggm && L->g_aggx && L->g_P && L->g_xrow && L->grads && L->wpack,
             "edge_backward: null buffer");
  const bool det = has(L, FASTEGNN_F_DETERMINISTIC);   // store + CSC reduce instead of the atomic scatter
  FE_REQUIRE(det ? L->g_QXe != nullptr : L->g_QX_src != nullptr, "edge_backward: g_QXe (deterministic) / g_QX_src null");
  const fastegnn_graph_t &gr = L->graph;
  FE_REQUIRE(!(det && has(L, FASTEGNN_F_GQX_ACCUM)), "edge_backward: FASTEGNN_F_GQX_ACCUM needs the atomic scatter");
  if (!det && gr.n_src > 0 && !has(L, FASTEGNN_F_GQX_ACCUM))
    (void)hipMemsetAsync(L->g_QX_src, 0, (size_t)gr.n_src * QXLD * sizeof(float), st);
  if (gr.n_edges == 0 || L->N == 0) {   // nothing to walk (with edges the kernel writes every row of g_P / g_xrow)
    (void)hipMemsetAsync(L->g_P, 0, (size_t)L->N * H * sizeof(float), st);
    (void)hipMemsetAsync(L->g_xrow, 0, (size_t)L->N * 3 * sizeof(float), st);
    return check_launch("edge_backward(memset)");
  }
  float *const *g = L->grads;
  EdgeBwdArgs A;
  A.f = make_edge_args(L);
  A.g_aggm = L->g_aggm; A.g_aggx = L->g_aggx; A.g_P = L->g_P; A.g_xrow = L->g_xrow; A.g_QXe = L->g_QXe;
  A.g_QXs_atomic = det ? nullptr : L->g_QX_src;
  A.g_ea = L->ea > 0 ? L->g_ea_sorted : nullptr;
  A.ld_e0 = 2 * H + 1 + L->ea;
  A.d_wr = g[FASTEGNN_P_EDGE0_W] + (has(L, FASTEGNN_F_EGNN) ? 0 : 2 * H);   // basic.py:313: radial is column 0
  A.d_we = g[FASTEGNN_P_EDGE0_W] + 2 * H + 1;
  A.d_bx2 = g[FASTEGNN_P_CR2_B];
  A.d_wx2 = g[FASTEGNN_P_CR2_W]; A.d_attw = g[FASTEGNN_P_ATT_W]; A.d_attb = g[FASTEGNN_P_ATT_B];
  FE_REQUIRE(!has(L, FASTEGNN_F_ATTENTION) || (A.d_attw && A.d_attb), "edge_backward: attention grads null");
  A.C = L->C;
  FE_REQUIRE((size_t)L->N * QXLD < (1u << 30) && (size_t)gr.n_src * QXLD < (1u << 30) && (size_t)gr.n_edges * 8 < (1u << 30),
             "edge_backward: tables exceed the 32-bit offset range of the gather path");
  {
    int grid = cdiv(gr.n_chunks, PC_PROD);   // small graphs: one 32-edge row chunk per producer wave (see edge_forward)
    if (grid > 256) grid = 256;
    WgradBatch local(L->wg_slab, st);
    WgradBatch &wb = shared ? *shared : local;
    int rc;
    if ((rc = wb.add_slabs(g[FASTEGNN_P_EDGE2_W], H, 0, 1, g[FASTEGNN_P_EDGE2_B], grid, &A.slab_w2))) return rc;
    if ((rc = wb.add_slabs(g[FASTEGNN_P_CR0_W], H, 0, 1, g[FASTEGNN_P_CR0_B], grid, &A.slab_wx1))) return rc;
    A.slab = wb.tab.slab;
    A.slab_b = wb.tab.slab_b;
    FE_REQUIRE(L->wg_edge, "edge_backward: wg_edge null");
    A.cons_scratch = L->wg_edge;
    const int imgf = has(L, FASTEGNN_F_BF16) ? pc_img_floats<GM_BF16>() : pc_img_floats<GM_EDGE_BWD>();
    const size_t lds = (imgf + EV_COUNT * H + PC_PROD * 16 * TS + 2 * PC_RING * PC_SLOT + PC_CTRL) * sizeof(float);
    {
      ProfScope _ps_edge_bwd_kernel(K_EDGE_BWD, st);
      const dim3 g3(grid), b3(64 * PC_WAVES);
      if (L->ea <= 2) {
        if (has(L, FASTEGNN_F_BF16)) hipLaunchKernelGGL((edge_bwd_pc_kernel<GM_BF16, 2>), g3, b3, lds, st, A);
        else hipLaunchKernelGGL((edge_bwd_pc_kernel<GM_EDGE_BWD, 2>), g3, b3, lds, st, A);
      } else {
        if (has(L, FASTEGNN_F_BF16)) hipLaunchKernelGGL((edge_bwd_pc_kernel<GM_BF16, 7>), g3, b3, lds, st, A);
        else hipLaunchKernelGGL((edge_bwd_pc_kernel<GM_EDGE_BWD, 7>), g3, b3, lds, st, A);
      }
    }
    if ((rc = check_launch("edge_bwd_pc_kernel"))) return rc;
    return shared ? FASTEGNN_OK : wb.finish();
  }
}

}  // namespace fe
// the edge stage contracts its weight gradients inside the workgroup: no operand workspace, only the running sums of its
// two consumer waves per workgroup (256 x 2 tiles of 64x64)
extern "C" size_t fastegnn_wg_edge_floats(int32_t E) { (void)E; return (size_t)256 * 2 * fe::IMG; }
// weight-gradient operand workspaces of the virtual / node-level stages (layouts: virt_backward in virt_bwd.hip,
// graph_post_backward, graph_pre_backward, node_pre_backward above).  The flag-less query is the upper bound over both forms of B4
// (the tile-major form: v / Gv + the per-group parts of g_A / g_x + consumer scratch); the _for variant sizes by the form that runs.
extern "C" size_t fastegnn_wg_virt_floats(int32_t N, int32_t C) {
  // the upper bound over BOTH forms of B4: which one runs depends on N, C, the flags and the FASTEGNN_VIRT_CS* switches, and the phased
  // form's workspace (256 (5 + C) tiles, independent of N) exceeds the tile-major one on small inputs
  const size_t pc = C >= 1 ? fe::virt_pc_wg_floats((size_t)(N > 0 ? N : 0), (size_t)C) : 0;
  const size_t cs = C >= 1 ? fe::virt_cs_wg_floats((size_t)C) : 0;
  const size_t n = pc > cs ? pc : cs;
  return n > 4 ? n : 4;
}
extern "C" size_t fastegnn_wg_virt_floats_for(int32_t N, int32_t C, int32_t flags) {
  // the channel-phased form of B4 (round 5) keeps no [C][N][64] array: consumer scratch + the partial slabs of dW3c only
  if (fe::virt_cs_applies(N, C, flags)) return fe::virt_cs_wg_floats((size_t)C);
  const size_t n = C >= 1 ? fe::virt_pc_wg_floats((size_t)(N > 0 ? N : 0), (size_t)C) : 0;
  return n > 4 ? n : 4;
}
extern "C" size_t fastegnn_wg_node_floats(int32_t N, int32_t B, int32_t C) {
  const size_t m = (size_t)(N > 0 ? N : 0), g = (size_t)(B > 0 ? B : 0) * (size_t)(C > 0 ? C : 0);
  return 8 * (m > g ? m : g) * fe::H + 4;
}
// floats of ALL backward scratch arrays of fastegnn_layer_t (g_poolV .. wg_slab), each rounded up to a multiple of 4
// floats (16-byte aligned carving of one allocation)
static size_t scratch_floats(int32_t N, int32_t E, int32_t n_src, int32_t B, int32_t C, size_t wg_virt, bool per_edge_rows = true) {
  auto r4 = [](size_t n) { return (n + 3) / 4 * 4; };
  const size_t n = N > 0 ? N : 0, e = E > 0 ? E : 1, s = n_src > 0 ? n_src : 0, bc = (size_t)(B > 0 ? B : 0) * (size_t)(C > 0 ? C : 0);
  size_t t = 0;
  t += 2 * r4(bc * fe::H) + 2 * r4(bc * 3) + r4((size_t)B * 4);          // g_poolV g_Bc | g_poolX g_Zp | g_xbar
  t += 3 * r4(n * fe::H) + 2 * r4(n * 3) + 2 * r4(n);                    // g_A g_P g_aggm | g_aggx g_xrow | g_svel g_sgrav
  t += (per_edge_rows ? r4(e * fe::QXLD) : 4) + r4(s * fe::QXLD);        // g_QXe (FASTEGNN_F_DETERMINISTIC only) | g_QX_src
  t += r4(fastegnn_wg_edge_floats(E)) + r4(wg_virt) + r4(fastegnn_wg_node_floats(N, B, C)) + r4(fastegnn_wg_slab_floats());
  return t;
}
extern "C" size_t fastegnn_backward_scratch_floats(int32_t N, int32_t E, int32_t n_src, int32_t B, int32_t C) {
  return scratch_floats(N, E, n_src, B, C, fastegnn_wg_virt_floats(N, C));
}
extern "C" size_t fastegnn_backward_scratch_floats_for(int32_t N, int32_t E, int32_t n_src, int32_t B, int32_t C, int32_t flags) {
  return scratch_floats(N, E, n_src, B, C, fastegnn_wg_virt_floats_for(N, C, flags), (flags & FASTEGNN_F_DETERMINISTIC) != 0);
}
namespace fe {

// B2b: col-keyed reduction of the per-edge d/d(Q|x) rows into the source table
__global__ __launch_bounds__(256) void edge_col_reduce_kernel(const float *g_QXe, const int32_t *cscptr,
                                                              const int32_t *csc_eid, int n_src, float *g_QXs) {
  const int l = lane_id();
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int n = wave; n < n_src; n += nwaves) {
    const int s = cscptr[n], e = cscptr[n + 1];
    float acc = 0.f, accx = 0.f;
    int k = s;
    for (; k + 4 <= e; k += 4) {   // four indexed rows in flight
      const float *r0 = g_QXe + (size_t)csc_eid[k] * QXLD, *r1 = g_QXe + (size_t)csc_eid[k + 1] * QXLD;
      const float *r2 = g_QXe + (size_t)csc_eid[k + 2] * QXLD, *r3 = g_QXe + (size_t)csc_eid[k + 3] * QXLD;
      const float a0 = r0[l], a1 = r1[l], a2 = r2[l], a3 = r3[l];
      float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
      if (l < 4) { x0 = r0[H + l]; x1 = r1[H + l]; x2 = r2[H + l]; x3 = r3[H + l]; }
      acc += a0; acc += a1; acc += a2; acc += a3;        // fixed order: deterministic
      accx += x0; accx += x1; accx += x2; accx += x3;
    }
    for (; k < e; ++k) {
      const float *row = g_QXe + (size_t)csc_eid[k] * QXLD;
      acc += row[l];
      if (l < 4) accx += row[H + l];
    }
    g_QXs[(size_t)n * QXLD + l] = acc;
    if (l < 4) g_QXs[(size_t)n * QXLD + H + l] = accx;
  }
}
int edge_col_reduce(const fastegnn_layer_t *L, hipStream_t st) {
  const fastegnn_graph_t &gr = L->graph;
  if (!has(L, FASTEGNN_F_DETERMINISTIC)) return FASTEGNN_OK;   // edge_backward has zeroed g_QX_src and scattered into it
  FE_REQUIRE(L->g_QX_src && (gr.n_edges == 0 || (L->g_QXe && gr.cscptr && gr.csc_eid)), "edge_col_reduce: null buffer");
  if (gr.n_src == 0) return FASTEGNN_OK;
  if (gr.n_edges == 0) {
    (void)hipMemsetAsync(L->g_QX_src, 0, (size_t)gr.n_src * QXLD * sizeof(float), st);
    return check_launch("edge_col_reduce(memset)");
  }
  int grid = cdiv(gr.n_src, 4);
  if (grid > 2048) grid = 2048;
  { ProfScope _ps_edge_col_reduce_kernel(K_COL_REDUCE, st); hipLaunchKernelGGL(edge_col_reduce_kernel, dim3(grid), dim3(256), 0, st, L->g_QXe, gr.cscptr, gr.csc_eid, gr.n_src,
                     L->g_QX_src); }
  return check_launch("edge_col_reduce_kernel");
}

// =====================================================================================
// B1 node_pre_bwd: adjoint of S1, completes g_h / g_x / g_vel of the layer inputs
// =====================================================================================
struct NodePreBwdArgs {
  const float *h, *wpack, *g_P, *g_QX, *g_A, *g_svel, *g_sgrav, *g_xrow, *g_xbar, *g_x_out, *svel;
  const float *bv0, *wv2, *bg0, *wg2;
  const int32_t *batch;
  float *g_h, *g_x, *g_vel, *wg_gzv, *wg_gzg;
  float *d_wv2, *d_bv2, *d_wg2, *d_bg2;
  int N, gravity, has_vel;
  const float *vel, *wv0;   // FastRF velocity head: coord_mlp_vel(||vel||), wv0 = coord_mlp_vel.0.weight [H,1]
  float *d_wv0, *d_bv0;
  int C;
  int flags = 0;
  float act_param = 0.f;
};
// MODE: GM_X3 / GM_BF16 as in node_pre_fwd_kernel.  LDS: three split images W1AT W1BT V1AT (transposed products only) and the
// ROW-MAJOR split images of WVEL0 and WG0 (common.h), each serving its product and its transpose: 127 KB instead of the
// 168 KB seven split images would take.
template <int MODE>
__global__ __launch_bounds__(64 * NODE_PRE_WAVES) void node_pre_bwd_kernel(NodePreBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float limg[];
  unsigned *img3 = reinterpret_cast<unsigned *>(limg);                    // W1AT W1BT V1AT (consecutive ids)
  char *rmv = reinterpret_cast<char *>(limg + 3 * IMG3), *rmg = rmv + RM_BYTES;   // WVEL0, WG0 (consecutive row-major slots)
  load_images_x3(img3, wpack_x3(a.wpack, a.C, I_W1AT), 3);
  if (a.has_vel || a.gravity) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(wpack_rm(a.wpack, a.C, rm_slot(I_WVEL0)));
    u32x4 *dst = reinterpret_cast<u32x4 *>(rmv);
    for (int i = threadIdx.x; i < 2 * RM_BYTES / 16; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  typedef typename OperandOf<MODE>::type Op;
  auto op = [](const Vec &v) -> Op { return make_operand<MODE>(v); };
  const int l = lane_id(), j = l & 15, q = l >> 4;
  const int wave = global_wave_id(), nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ntiles = (a.N + 15) >> 4;
  Vec acc_wv2 = vzero(), acc_wg2 = vzero(), acc_wv0 = vzero(), acc_bv0 = vzero();
  // the two scalar head biases: sums of N per-node values with cancellation; double accumulators (one add per tile) keep
  // them at the reference's level (a float chain + float atomics measured 1.07e-6 against a reference at 3e-8)
  double acc_bv2 = 0.0, acc_bg2 = 0.0;
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int n = tile * 16 + j;
    const bool valid = n < a.N;
    const int nc = valid ? n : a.N - 1;
    Vec g_h = vload_row(a.g_h + (size_t)nc * H, q);
    gemm_op<MODE>(img3, 0, op(vload_row(a.g_P + (size_t)nc * H, q)), g_h);
    gemm_op<MODE>(img3, 1, op(vload_row(a.g_QX + (size_t)nc * QXLD, q)), g_h);
    gemm_op<MODE>(img3, 2, op(vload_row(a.g_A + (size_t)nc * H, q)), g_h);
    if (a.has_vel || a.gravity) {
      const Op hv = op(vload_row(a.h + (size_t)nc * H, q));   // one split / rounding feeds both heads
      if (a.has_vel) {  // coord_mlp_vel head (:139)
        Vec z = vload_vec(a.bv0, q);
        gemm_rm<MODE, false>(rmv, hv, z);
        const float gs = valid ? a.g_svel[nc] : 0.f;
        vaxpy(acc_wv2, gs, vsilu(z FE_ACT(a)));
        if (q == 0) acc_bv2 += gs;
        const Vec g_z = vdsilu_mul(vscale(vload_vec(a.wv2, q), gs), z FE_ACT(a));
        if (valid) vstore_row(a.wg_gzv + (size_t)n * H, q, g_z);
        gemm_rm<MODE, true>(rmv, op(g_z), g_h);
      }
      if (a.gravity) {  // gravity_mlp head (:142)
        Vec z = vload_vec(a.bg0, q);
        gemm_rm<MODE, false>(rmg, hv, z);
        const float gs = valid ? a.g_sgrav[nc] : 0.f;
        vaxpy(acc_wg2, gs, vsilu(z FE_ACT(a)));
        if (q == 0) acc_bg2 += gs;
        const Vec g_z = vdsilu_mul(vscale(vload_vec(a.wg2, q), gs), z FE_ACT(a));
        if (valid) vstore_row(a.wg_gzg + (size_t)n * H, q, g_z);
        gemm_rm<MODE, true>(rmg, op(g_z), g_h);
      }
    }
    if (a.wv0) {  // FastRF.py:139: only parameter gradients (the norm of the velocity is detached)
      const float vx = a.vel[(size_t)nc * 3], vy = a.vel[(size_t)nc * 3 + 1], vz = a.vel[(size_t)nc * 3 + 2];
      const float vn = sqrt_f(vx * vx + vy * vy + vz * vz);
      Vec z = vload_vec(a.bv0, q);
      vaxpy(z, vn, vload_vec(a.wv0, q));
      const float gs = valid ? a.g_svel[nc] : 0.f;
      vaxpy(acc_wv2, gs, vsilu(z FE_ACT(a)));
      if (q == 0) acc_bv2 += gs;
      const Vec g_z = vdsilu_mul(vscale(vload_vec(a.wv2, q), gs), z FE_ACT(a));
      vadd(acc_bv0, g_z);
      vaxpy(acc_wv0, vn, g_z);
    }
    if (valid) {
      vstore_row(a.g_h + (size_t)n * H, q, g_h);
      if (q == 0) {
        const int b = a.batch[n];
        const float sv = a.svel[n];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          a.g_x[(size_t)n * 3 + k] += a.g_xrow[(size_t)n * 3 + k] + a.g_QX[(size_t)n * QXLD + H + k] + a.g_xbar[b * 3 + k];
          if (a.g_vel) a.g_vel[(size_t)n * 3 + k] += sv * a.g_x_out[(size_t)n * 3 + k];
        }
      }
    }
  }
  __shared__ float red[4 * H + 2];
  __shared__ double red_d[2];
  for (int i = threadIdx.x; i < 4 * H + 2; i += blockDim.x) red[i] = 0.f;
  if (threadIdx.x < 2) red_d[threadIdx.x] = 0.0;
  __syncthreads();
  auto dsum = [&](double v) {   // sum over the 64 lanes of the wave
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
  };
  double s = 0.0;
  if (a.has_vel || a.wv0) {
    vec_reduce_lds(red, acc_wv2, j, q);
    s = dsum(acc_bv2);
    if (l == 0) atomicAdd(&red_d[0], s);
  }
  if (a.wv0) {
    vec_reduce_lds(red + 2 * H + 2, acc_wv0, j, q);
    vec_reduce_lds(red + 3 * H + 2, acc_bv0, j, q);
  }
  if (a.gravity) {
    vec_reduce_lds(red + H, acc_wg2, j, q);
    s = dsum(acc_bg2);
    if (l == 0) atomicAdd(&red_d[1], s);
  }
  __syncthreads();
  if (threadIdx.x < H) {
    if (a.has_vel || a.wv0) {
      atomicAdd(&a.d_wv2[threadIdx.x], red[threadIdx.x]);
      if (threadIdx.x == 0) atomicAdd(a.d_bv2, (float)red_d[0]);
    }
    if (a.wv0) {
      atomicAdd(&a.d_wv0[threadIdx.x], red[2 * H + 2 + threadIdx.x]);
      atomicAdd(&a.d_bv0[threadIdx.x], red[3 * H + 2 + threadIdx.x]);
    }
    if (a.gravity) {
      atomicAdd(&a.d_wg2[threadIdx.x], red[H + threadIdx.x]);
      if (threadIdx.x == 0) atomicAdd(a.d_bg2, (float)red_d[1]);
    }
  }
}

int node_pre_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared) {
  FE_REQUIRE(L->h && L->wpack && L->g_P && L->g_QX && L->g_A && L->g_svel && L->g_xrow && L->g_xbar && L->g_x_out &&
                 L->svel && L->g_h && L->g_x && L->wg_node && L->grads && L->batch,
             "node_pre_backward: null buffer");
  if (L->N == 0) return FASTEGNN_OK;
  const bool grav = has(L, FASTEGNN_F_GRAVITY), rf = has(L, FASTEGNN_F_RF);
  const float *const *p = L->params;
  float *const *g = L->grads;
  const int N = L->N;
  float *wg_np = L->wg_node + 2 * wg_node_rows(L) * H;   // this stage's operand region of wg_node (kernels.h)
  float *wg_gzv = wg_np, *wg_gzg = wg_np + (size_t)N * H;
  NodePreBwdArgs a{L->h, L->wpack, L->g_P, L->g_QX, L->g_A, L->g_svel, L->g_sgrav, L->g_xrow, L->g_xbar, L->g_x_out,
                   L->svel, p[FASTEGNN_P_VEL0_B], p[FASTEGNN_P_VEL2_W], p[FASTEGNN_P_GRAV0_B], p[FASTEGNN_P_GRAV2_W],
                   L->batch, L->g_h, L->g_x, L->g_vel, wg_gzv, wg_gzg,
                   g[FASTEGNN_P_VEL2_W], g[FASTEGNN_P_VEL2_B], g[FASTEGNN_P_GRAV2_W], g[FASTEGNN_P_GRAV2_B], N, grav ? 1 : 0,
                   (p[FASTEGNN_P_VEL0_W] && !rf) ? 1 : 0, L->vel, rf ? p[FASTEGNN_P_VEL0_W] : nullptr,
                   g[FASTEGNN_P_VEL0_W], g[FASTEGNN_P_VEL0_B], L->C, L->flags, L->act_param};
  FE_REQUIRE(!rf || (L->vel && p[FASTEGNN_P_VEL0_W] && g[FASTEGNN_P_VEL0_W] && g[FASTEGNN_P_VEL0_B] && g[FASTEGNN_P_VEL2_W] &&
                     g[FASTEGNN_P_VEL2_B]),
             "node_pre_backward: FastRF needs vel and the coord_mlp_vel parameters / gradients");
  int grid = cdiv(cdiv(N, 16), NODE_PRE_WAVES);
  if (grid > 256) grid = 256;
  {
    ProfScope _ps_node_pre_bwd_kernel(K_NODE_PRE_BWD, st);
    const size_t lds = (size_t)3 * IMG3 * sizeof(float) + 2 * RM_BYTES;
    if (has(L, FASTEGNN_F_BF16)) hipLaunchKernelGGL(node_pre_bwd_kernel<GM_BF16>, dim3(grid), dim3(64 * NODE_PRE_WAVES), lds, st, a);
    else hipLaunchKernelGGL(node_pre_bwd_kernel<GM_X3>, dim3(grid), dim3(64 * NODE_PRE_WAVES), lds, st, a);
  }
  int rc = check_launch("node_pre_bwd_kernel");
  if (rc) return rc;
  const int ld_e0 = 2 * H + 1 + L->ea, ld_v0 = 2 * H + 1 + L->C;
  // edge_mlp.0 columns [0,H) <- h[row] (P), [H,2H) <- h[col] (Q), bias through P
  WgradBatch local(L->wg_slab, st);
  WgradBatch &wb = shared ? *shared : local;
  wb.round = has(L, FASTEGNN_F_BF16);
  const int c0 = has(L, FASTEGNN_F_EGNN) ? 1 : 0;   // EGNN baseline: [radial | h_row | h_col | edge_attr]
  if ((rc = wb.add(L->g_P, H, L->h, H, N, g[FASTEGNN_P_EDGE0_W], ld_e0, c0, 1, g[FASTEGNN_P_EDGE0_B]))) return rc;
  if ((rc = wb.add(L->g_QX, QXLD, L->h, H, N, g[FASTEGNN_P_EDGE0_W], ld_e0, c0 + H, 1, nullptr))) return rc;
  // edge_mlp_virtual.0 columns [0,H) <- h (A)   (absent for the EGNN baseline: add() skips a null dW)
  if ((rc = wb.add(L->g_A, H, L->h, H, N, g[FASTEGNN_P_VIRT0_W], ld_v0, 0, 1, nullptr))) return rc;
  if (p[FASTEGNN_P_VEL0_W] && !rf)
    if ((rc = wb.add(wg_gzv, H, L->h, H, N, g[FASTEGNN_P_VEL0_W], H, 0, 1, g[FASTEGNN_P_VEL0_B]))) return rc;
  if (grav)
    if ((rc = wb.add(wg_gzg, H, L->h, H, N, g[FASTEGNN_P_GRAV0_W], H, 0, 1, g[FASTEGNN_P_GRAV0_B]))) return rc;
  return shared ? FASTEGNN_OK : wb.finish();
}

}  // namespace fe


#ifdef FE_STAMP
extern "C" int fastegnn_debug_read_eb_stamps(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fe::g_stamps), z, sizeof(z));
  }
  return 0;
}
#endif
