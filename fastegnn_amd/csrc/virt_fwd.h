// S4 virt: the virtual-stage forward kernel (reference: models/FastEGNN.py:111-119,136-166; csrc/layer_fwd.hip holds its launcher and the
// section comment).  In a header since round 6: the two-waves-per-tile instantiation (PAIR) is compiled in its own translation unit,
// virt_fwd_pair.hip, WITHOUT -amdgpu-sched-strategy=max-memory-clause -- with that option this hipcc's backend dies on it ("Illegal
// instruction detected: Operand has incorrect register class.  V_CMP_NE_U32_e32 0, $src_shared_base"), with the default scheduler it builds.
#pragma once
#include "stages.h"

namespace fe {

constexpr int VIRT_FWD_IMG_FLOATS = 5 * IMG3;
// PAIR (round 6, small inputs): TWO waves per tile -- waves w and w + 4 take the even / the odd channels of tile w & 3 --, so a step is
// four tiles and half the serial chain: a 16-channel tile costs ~100 us whatever N is, and below 8 x 256 tiles most waves of the chip had
// nothing to do (a 12 500-node shard of the cfg4 frame, the N-body mini-batches).  The stage then holds the images of two channels per
// iteration, four f16x2 images of two parts (4 096 words) each; the two waves' shares of the node-MLP accumulator and of the coordinate
// update meet in LDS before the tile's tail.  f16x2 build only.
constexpr int VIRT_FWD_PAIR_STAGE = 4 * 4096;
constexpr int VIRT_FWD_IMG_FLOATS_PAIR = 3 * IMG3 + VIRT_FWD_PAIR_STAGE;
inline size_t virt_fwd_lds_bytes(int C, bool pair = false) {
  return (size_t)((pair ? VIRT_FWD_IMG_FLOATS_PAIR : VIRT_FWD_IMG_FLOATS) + VV_COUNT * H + 2 * (C * H + ((3 * C + 3) & ~3)) + 4) * sizeof(float);   // + 4 control words
}
template <int MODE, bool PAIR = false>
__global__ __launch_bounds__(64 * VIRT_WAVES) void virt_fwd_kernel(VirtArgs a) {
  static_assert(!PAIR || MODE == GM_F16, "the two-waves-per-tile walk is built for the f16x2 images");
  constexpr int TPS = PAIR ? VIRT_WAVES / 2 : VIRT_WAVES;   // tiles per step of a workgroup
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = a.C;
  float *img = lds;                              // V2, WXV0, WXX0 (split images)
  unsigned *stage = reinterpret_cast<unsigned *>(lds) + 3 * IMG3;   // W3c[c] of the channel in flight (slot c & 1) and of the next
  float *vec = lds + (PAIR ? VIRT_FWD_IMG_FLOATS_PAIR : VIRT_FWD_IMG_FLOATS);        // VV_COUNT vectors
  float *poolV_l = vec + VV_COUNT * H;           // [C][64]
  float *poolX_l = poolV_l + C * H;              // [3][C]
  float *Bc_l = poolX_l + ((3 * C + 3) & ~3);    // [C][64]: Bc rows of the graph the workgroup is in
  float *Z_l = Bc_l + C * H;                     // [3][C]: its virtual coordinates
#ifdef VF_SPLIT_BARRIER
  // measured alternative to the per-channel workgroup barrier: per stage slot a count of committed image shares and a count
  // of waves that are done with the image (monotonic; a wave waits only for what it needs, at most one channel of skew)
  int *ctl = reinterpret_cast<int *>(Z_l + ((3 * C + 3) & ~3));   // filled[2] | done[2]
  if (threadIdx.x < 4) ctl[threadIdx.x] = 0;
  int vf_steps = 0;   // staged steps this workgroup has completed (the same in every wave)
  auto vf_signal = [&](int k) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane_id() == 0) atomicAdd(&ctl[k], 1);
  };
  auto vf_wait = [&](int k, int need) {
    while (__atomic_load_n(&ctl[k], __ATOMIC_RELAXED) < need) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
#endif
  load_images_x3(reinterpret_cast<unsigned *>(img), wpack_x3(a.wpack, C, I_V2), 3);
  virt_load_vecs(vec, a);
  for (int i = threadIdx.x; i < C * H + 3 * C; i += blockDim.x) poolV_l[i] = 0.f;
  __syncthreads();
  const int l = lane_id(), j = l & 15, q = l >> 4, wv = wave_id();
  // a workgroup owns a contiguous run of 16-node tiles and walks it VIRT_WAVES tiles at a time; the runs differ by
  // at most one tile, so the last, partial step of a workgroup is a single wave that has its SIMD to itself
  const int ntiles = (a.N + 15) >> 4;
  const int t_lo = (int)((long)blockIdx.x * ntiles / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x);
  const float invC = C > 0 ? 1.0f / (float)C : 0.f;
  const bool clamp_aggx = a.flags & FASTEGNN_F_EGNN;   // basic.py:310
  const bool rf = a.flags & FASTEGNN_F_RF;             // FastRF.py:155-186: no node_model / node_model_virtual
  int cur = -1;  // graph the LDS pool accumulators belong to
  VF_T0()
  auto flush_pools = [&]() {
    if (!rf)
      for (int i = threadIdx.x; i < C * H; i += blockDim.x) {
        atomicAdd(&a.poolV[(size_t)cur * C * H + i], poolV_l[i]);
        poolV_l[i] = 0.f;
      }
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) {
      atomicAdd(&a.poolX[(size_t)cur * 3 * C + i], poolX_l[i]);
      poolX_l[i] = 0.f;
    }
  };
  // the K = H*C contraction of node_mlp.0 reads W3c[c] from an LDS stage of two slots that the whole workgroup refills (all
  // waves walk the channels in step): channel c + 1 is written into the other slot while channel c is in use, channel
  // c + 2 is on its way into registers -- ONE workgroup barrier per channel.
  // 16-byte pieces per thread: one per part of the image (h | m | l; an f16x2 image has two parts, the third is not copied)
  static_assert(2048 == 4 * 64 * VIRT_WAVES, "one part of a split image per pass of the stage copy");
  constexpr int STG = MODE == GM_F16 ? 2 : 3;
  u32x4 pre_w[PAIR ? 2 * STG : STG];
  // (PAIR: the argument is the ITERATION; iteration `it` needs the images of channels 2 it and 2 it + 1, in slots 2 (it & 1) + {0, 1})
  auto fetch_w3c = [&](int c) {
    if constexpr (PAIR) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cc = min(2 * c + g, C - 1);   // (an odd C: the last iteration's second image is a copy nobody reads)
        const u32x4 *src = reinterpret_cast<const u32x4 *>(wpack_x3(a.wpack, C, img_w3c(cc)));
#pragma unroll
        for (int i = 0; i < STG; ++i) pre_w[g * STG + i] = src[threadIdx.x + i * 64 * VIRT_WAVES];
      }
    } else {
      const u32x4 *src = reinterpret_cast<const u32x4 *>(wpack_x3(a.wpack, C, img_w3c(c)));
#pragma unroll
      for (int i = 0; i < STG; ++i) pre_w[i] = src[threadIdx.x + i * 64 * VIRT_WAVES];
    }
  };
  auto commit_w3c = [&](int slot) {
    if constexpr (PAIR) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(stage + (2 * slot + g) * 4096);
#pragma unroll
        for (int i = 0; i < STG; ++i) dst[threadIdx.x + i * 64 * VIRT_WAVES] = pre_w[g * STG + i];
      }
    } else {
      u32x4 *dst = reinterpret_cast<u32x4 *>(stage + slot * IMG3);
#pragma unroll
      for (int i = 0; i < STG; ++i) dst[threadIdx.x + i * 64 * VIRT_WAVES] = pre_w[i];
    }
  };
  const int grp = PAIR ? wv >> 2 : 0;            // PAIR: which of the tile's two waves this is (channels grp, grp + 2, ...)
  for (int tb = t_lo; tb < t_hi; tb += TPS) {
    // a single left-over tile is dealt to the waves by channel (wave w takes c = w, w + VIRT_WAVES, ...); the
    // channel sums of the node-MLP accumulator and of the coordinate update are combined through LDS.
    // (Round 4 measured the same deal for EVERY tile of a small shard -- 3 tiles per workgroup at 12 500 nodes, an emulated rank of
    // eight: 0.517 against 0.479 ms per step; the per-tile epilogue through the image stage costs what the idle waves cost.)
    const bool split = !PAIR && t_hi - tb == 1 && C >= VIRT_WAVES;
    const bool own = PAIR ? grp == 0 : (!split || wv == 0);
    const int n0 = tb * 16, nend = min(a.N, min(t_hi, tb + TPS) * 16);
    const int bfirst = a.batch[n0], blast = a.batch[nend - 1];
    const bool fast = bfirst == blast;   // every node of this step in ONE graph: its pools, Bc rows and Z sit in LDS
    const bool staged = C > 0 && !rf && !split;
    if (staged) fetch_w3c(0);
    if (fast && bfirst != cur) {
      __syncthreads();
      if (cur >= 0) flush_pools();
      cur = bfirst;
      for (int i = threadIdx.x; i < C * H; i += blockDim.x) Bc_l[i] = a.Bc[(size_t)cur * C * H + i];
      for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) Z_l[i] = a.Z[(size_t)cur * 3 * C + i];
      __syncthreads();
    }
    const int nb = split ? n0 : n0 + (PAIR ? (wv & 3) : wv) * 16;
    const int nvalid = max(0, min(16, nend - nb));
    const bool active = nvalid > 0;
    const int n = nb + j;
    const bool valid = n < nend;
    const int nc = valid ? n : nend - 1;
    const int b = a.batch[nc];
    Vec Ai;
    float xi[3] = {0.f, 0.f, 0.f};
    if (active) {
      Ai = vload_row(a.A + (size_t)nc * H, q);
      xi[0] = a.x[(size_t)nc * 3]; xi[1] = a.x[(size_t)nc * 3 + 1]; xi[2] = a.x[(size_t)nc * 3 + 2];
    }
    float transv[3] = {0.f, 0.f, 0.f};
    Vec nodeacc = own ? vload_vec(vec + VV_B3 * H, q) : vzero();
    if (staged) {
      __syncthreads();            // every wave is done with the stage (the previous step's node-level images)
      commit_w3c(0);
#ifdef VF_SPLIT_BARRIER
      vf_signal(0);
#endif
      if ((PAIR ? (C + 1) / 2 : C) > 1) fetch_w3c(1);
    }
    // iterations of the channel walk: C (a wave per tile), ceil(C / 2) (PAIR), this wave's share of C (a left-over tile dealt by channel)
    const int n_it = PAIR ? (C + 1) / 2 : (split ? (C - wv + VIRT_WAVES - 1) / VIRT_WAVES : C);
    VF_T(8)   // tile head: row loads, bookkeeping
    for (int it = 0; it < n_it; ++it) {
      const int c = PAIR ? 2 * it + grp : (split ? wv + it * VIRT_WAVES : it);
      const bool chan = !PAIR || c < C;          // (PAIR, odd C: the second wave sits out the last iteration)
#if defined(VF_SPLIT_BARRIER)
      static_assert(!PAIR, "VF_SPLIT_BARRIER is a lever of the one-wave-per-tile walk");
      if (staged) {
        const int sl = c & 1, s1 = sl ^ 1, per0 = (C + 1) >> 1, per1 = C >> 1;
        if (c + 1 < C) {
          // every wave is done with channel c - 1 (the previous image of slot s1) ...
          vf_wait(2 + s1, VIRT_WAVES * (vf_steps * (s1 ? per1 : per0) + ((c + 2 - s1) >> 1)));
          commit_w3c(s1);           // ... this wave's share of W3c[c + 1]
          vf_signal(s1);
          if (c + 2 < C) fetch_w3c(c + 2);
        }
        vf_wait(sl, VIRT_WAVES * (vf_steps * (sl ? per1 : per0) + (c >> 1) + 1));   // every share of W3c[c] is in slot c & 1
      }
#elif !defined(VF_DIAG_NOSTAGE)   // diagnostic: what do the per-channel stage refill and its barrier cost? (results are wrong without them)
      if (staged) {
        __syncthreads();          // W3c[c] is in slot c & 1; every wave is done with channel c - 1, i.e. with the other slot
        if (it + 1 < n_it) {
          commit_w3c((it + 1) & 1);
          if (it + 2 < n_it) fetch_w3c(it + 2);
        }
      }
#endif
      VF_T(0)   // per-channel barrier + stage refill
      if (active && chan) {
        VirtFwdState<MODE> S;
        virt_tile_forward<MODE>(a, img, vec, Ai, xi, b, c, q, fast ? Bc_l : nullptr, fast ? Z_l : nullptr, S VF_TA);
        transv[0] -= S.vd[0] * S.sx;
        transv[1] -= S.vd[1] * S.sx;
        transv[2] -= S.vd[2] * S.sx;
        // pools: sums over the nodes of the tile (transposing DPP butterfly, then one 64-lane atomic per accumulator)
        if (fast) {
          if (!rf) tile_sum_add(poolV_l + c * H, nvalid == 16 ? S.v : (valid ? S.v : vzero()), j, q);
          // (lane k < 3 adds component k: per-lane addresses, so the compiler's uniform-address atomic combiner stays out)
          const float p0 = jsum_dpp(valid ? S.vd[0] * S.sX : 0.f), p1 = jsum_dpp(valid ? S.vd[1] * S.sX : 0.f),
                      p2 = jsum_dpp(valid ? S.vd[2] * S.sX : 0.f);
          if (l < 3) atomicAdd(&poolX_l[l * C + c], l == 0 ? p0 : (l == 1 ? p1 : p2));
        } else {
          // the step spans graphs: straight to the global pools, one pass per graph present in this tile
          const int g_lo = __builtin_amdgcn_readlane(b, 0), g_hi = __builtin_amdgcn_readlane(b, nvalid - 1);
          for (int g = g_lo; g <= g_hi; ++g) {
            const bool in_g = valid && b == g;
            if (!rf) tile_sum_add(a.poolV + ((size_t)g * C + c) * H, in_g ? S.v : vzero(), j, q);
            const float p0 = jsum_dpp(in_g ? S.vd[0] * S.sX : 0.f), p1 = jsum_dpp(in_g ? S.vd[1] * S.sX : 0.f),
                        p2 = jsum_dpp(in_g ? S.vd[2] * S.sX : 0.f);
            if (l < 3) atomicAdd(&a.poolX[((size_t)g * 3 + l) * C + c], l == 0 ? p0 : (l == 1 ? p1 : p2));
          }
        }
        VF_T(6)   // pools
        if (!rf) {
#ifdef VF_DIAG_NODE_IMG0      // diagnostic: the node-MLP block product from a resident image instead of the stage (wrong results)
          gemm_op<MODE>(img, 0, S.vs, nodeacc);
#elif defined(VF_DIAG_NONODE)  // diagnostic: no node-MLP block product (wrong results)
          if (S.vs.p[0][0][0] == 0x12345678u) nodeacc = S.v;
#else
          if (split) gemm_op<MODE>(wpack_x3(a.wpack, C, img_w3c(c)), 0, S.vs, nodeacc);   // the stage serves the stepped walk only
          else if constexpr (PAIR) gemm64_f2<false>(stage, S.vs, nodeacc, (2 * (it & 1) + (int)(threadIdx.x >> 8)) * 1024);   // slot 2 (it & 1) + grp: 4 096 words each
          else gemm_op<MODE>(stage + (c & 1) * IMG3, 0, S.vs, nodeacc);
#endif
        }
        VF_T(7)   // node-MLP block product
      }
#ifdef VF_SPLIT_BARRIER
      if (staged) vf_signal(2 + (c & 1));   // this wave is done with the image of channel c
#endif
    }
#ifdef VF_SPLIT_BARRIER
    if (staged) ++vf_steps;
#endif
    if constexpr (PAIR) {   // the second wave's share of the tile's node-MLP accumulator and coordinate update -> the first wave
      float *comb = lds + 3 * IMG3;      // (the stage, as floats: four [16][TS] tiles)
      const int co = ((wv & 3) * 16 + j) * TS;
      __syncthreads();            // every wave is done with the stage
      if (grp == 1 && active) {
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(&comb[co + 16 * t + 4 * q]) = nodeacc.t[t];
        if (q == 0) *reinterpret_cast<f32x4 *>(&comb[co + H]) = f32x4{transv[0], transv[1], transv[2], 0.f};
      }
      __syncthreads();
      if (grp == 0 && active) {
#pragma unroll
        for (int t = 0; t < 4; ++t) nodeacc.t[t] += *reinterpret_cast<const f32x4 *>(&comb[co + 16 * t + 4 * q]);
#pragma unroll
        for (int k = 0; k < 3; ++k) transv[k] += comb[co + H + k];
      }
      // (the tail's first stage_image() opens with a barrier: the tiles are read before the stage is refilled)
    }
    if (split) {   // sum the waves' channel shares ([16][68] floats in the idle W3c stage)
      float *comb = reinterpret_cast<float *>(stage);
      __syncthreads();
      for (int i = threadIdx.x; i < 16 * TS; i += blockDim.x) comb[i] = 0.f;
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&comb[j * TS + 16 * t + 4 * q + r], nodeacc.t[t][r]);
      if (q == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&comb[j * TS + H + k], transv[k]);
      }
      __syncthreads();
      if (own) {
#pragma unroll
        for (int t = 0; t < 4; ++t) nodeacc.t[t] = *reinterpret_cast<const f32x4 *>(comb + j * TS + 16 * t + 4 * q);
#pragma unroll
        for (int k = 0; k < 3; ++k) transv[k] = comb[j * TS + H + k];
      }
      __syncthreads();
    }
    if (active && rf && own) {   // the node features pass through unchanged (FastRF.py:186)
      if (valid) vstore_row(a.h_out + (size_t)n * H, q, vload_row(a.h + (size_t)nc * H, q));
    }
    if (!rf) {
      // node_model: node_mlp.0 on [h | agg | flat(v) | node_attr]  (:153-166).  The three node-level images (W3A, W3B,
      // W4) pass through the idle W3c stage, one after the other -- every wave of the workgroup takes part in the copy,
      // the waves that own a tile run the products (read straight from global memory these three products cost more
      // than the whole channel loop of a tile, cf. the phase stamps of virt_bwd).
      const bool mine = active && own;
      // (round 5: the SPLIT images of the three weights -- f16x2 / bf16x3 / bf16 products on the matrix pipe like every other product of
      //  this kernel, where rounds 1-4 ran them as fp32-input MFMAs from the fp32 images: 3 x 64 MFMAs of 32 cycles per tile, a
      //  fifth of the kernel's matrix-pipe time, against 3 x 24 of 16)
      auto stage_image = [&](int id) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(wpack_x3(a.wpack, C, id));
        u32x4 tmp[STG];
#pragma unroll
        for (int i = 0; i < STG; ++i) tmp[i] = src[threadIdx.x + i * 64 * VIRT_WAVES];
        __syncthreads();          // every wave is done with the previous content of the stage
        u32x4 *dst = reinterpret_cast<u32x4 *>(stage);
#pragma unroll
        for (int i = 0; i < STG; ++i) dst[threadIdx.x + i * 64 * VIRT_WAVES] = tmp[i];
        __syncthreads();
      };
      Vec hv = vzero();
      stage_image(I_W3A);
      if (mine) {
        hv = vload_row(a.h + (size_t)nc * H, q);
        gemm_i<MODE>(stage, 0, hv, nodeacc);
      }
      stage_image(I_W3B);
      if (mine) {
        gemm_i<MODE>(stage, 0, vload_row(a.aggm + (size_t)nc * H, q), nodeacc);
        if (a.na > 0) {
          const int ld = 2 * H + H * C + a.na;
          for (int k = 0; k < a.na; ++k) {
            const float av = a.node_attr[(size_t)nc * a.na + k];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                nodeacc.t[t][r] += av * a.N0W[(size_t)(16 * t + 4 * q + r) * ld + 2 * H + H * C + k];
          }
        }
        if (valid) vstore_row(a.npre + (size_t)n * H, q, nodeacc);
      }
      stage_image(I_W4);
      if (mine) {
        Vec out = vload_vec(vec + VV_B4 * H, q);
        gemm_i<MODE>(stage, 0, vsilu(nodeacc FE_ACT(a)), out);
        if (a.flags & FASTEGNN_F_RESIDUAL) vadd(out, hv);
        if (valid) vstore_row(a.h_out + (size_t)n * H, q, out);
      }
    }
    VF_T(9)   // tile tail: node-level products through the stage
    if (active && own) {
      if (valid) {
        if (q == 0) {
          const float sv = a.svel[n];
          const float sg = (a.flags & FASTEGNN_F_GRAVITY) ? a.sgrav[n] : 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float ax = a.aggx[(size_t)n * 3 + k];
            if (clamp_aggx) ax = ax > 100.f ? 100.f : (ax < -100.f ? -100.f : ax);   // (NaN stays NaN, as torch.clamp: basic.py:310)
            a.x_out[(size_t)n * 3 + k] = xi[k] + ax + transv[k] * invC + sv * a.vel[(size_t)n * 3 + k] + sg * a.g[k];
          }
        }
      }
    }
  }
  __syncthreads();
  if (cur >= 0) flush_pools();
  VF_T(10)
  VF_TEND()
}

}  // namespace fe
