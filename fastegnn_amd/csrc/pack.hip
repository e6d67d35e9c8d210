// Weight packing: reference state_dict tensors ([out,in] row-major) -> MFMA operand images.
// One workgroup per 64x64 image; see common.h (img_index) for the layout.
#include "kernels.h"

namespace fe {

struct PackDesc {
  const float *src;  // null -> zero image
  int ld, c0, ks, transposed;
};

struct PackArgs {
  const float *p[FASTEGNN_P_COUNT];
  float *wpack;
  int C, ea, na, egnn, rf, bf16;
};
constexpr int PACK_MAX_LAYERS = 8;   // layers per launch of the multi-layer form (8 x 336 bytes of kernel arguments)
struct PackArgsN {
  PackArgs l[PACK_MAX_LAYERS];
};

__device__ __forceinline__ PackDesc pack_desc(const PackArgs &a, int id) {
  const int C = a.C;
  const int ld_e0 = 2 * H + 1 + a.ea, ld_v0 = 2 * H + 1 + C, ld_n0 = 2 * H + H * C + a.na;
  PackDesc d{nullptr, H, 0, 1, 0};
  auto set = [&](int slot, int ld, int c0, int tr) {
    d.src = a.p[slot];
    d.ld = ld;
    d.c0 = c0;
    d.transposed = tr;
  };
  switch (id) {
    case I_W2: set(FASTEGNN_P_EDGE2_W, H, 0, 0); break;
    case I_WX1: set(FASTEGNN_P_CR0_W, H, 0, 0); break;
    case I_W2T: set(FASTEGNN_P_EDGE2_W, H, 0, 1); break;
    case I_WX1T: set(FASTEGNN_P_CR0_W, H, 0, 1); break;
    case I_V2: set(FASTEGNN_P_VIRT2_W, H, 0, 0); break;
    case I_WXV0: set(FASTEGNN_P_CRV0_W, H, 0, 0); break;
    case I_WXX0: set(FASTEGNN_P_CVV0_W, H, 0, 0); break;
    case I_V2T: set(FASTEGNN_P_VIRT2_W, H, 0, 1); break;
    case I_WXV0T: set(FASTEGNN_P_CRV0_W, H, 0, 1); break;
    case I_WXX0T: set(FASTEGNN_P_CVV0_W, H, 0, 1); break;
    case I_W1A: set(FASTEGNN_P_EDGE0_W, ld_e0, a.egnn, 0); break;       // EGNN: column 0 is the radial
    case I_W1B: set(FASTEGNN_P_EDGE0_W, ld_e0, H + a.egnn, 0); break;
    case I_V1A: set(FASTEGNN_P_VIRT0_W, ld_v0, 0, 0); break;
    case I_WVEL0: if (!a.rf) set(FASTEGNN_P_VEL0_W, H, 0, 0); break;   // FastRF: [H,1] weight, no image
    case I_WG0: set(FASTEGNN_P_GRAV0_W, H, 0, 0); break;
    case I_W1AT: set(FASTEGNN_P_EDGE0_W, ld_e0, a.egnn, 1); break;
    case I_W1BT: set(FASTEGNN_P_EDGE0_W, ld_e0, H + a.egnn, 1); break;
    case I_V1AT: set(FASTEGNN_P_VIRT0_W, ld_v0, 0, 1); break;
    case I_WVEL0T: if (!a.rf) set(FASTEGNN_P_VEL0_W, H, 0, 1); break;
    case I_WG0T: set(FASTEGNN_P_GRAV0_W, H, 0, 1); break;
    case I_W5A: set(FASTEGNN_P_NODEV0_W, 2 * H, 0, 0); break;
    case I_W5B: set(FASTEGNN_P_NODEV0_W, 2 * H, H, 0); break;
    case I_W6: set(FASTEGNN_P_NODEV2_W, H, 0, 0); break;
    case I_W5AT: set(FASTEGNN_P_NODEV0_W, 2 * H, 0, 1); break;
    case I_W5BT: set(FASTEGNN_P_NODEV0_W, 2 * H, H, 1); break;
    case I_W6T: set(FASTEGNN_P_NODEV2_W, H, 0, 1); break;
    case I_V1B: set(FASTEGNN_P_VIRT0_W, ld_v0, H, 0); break;
    case I_V1BT: set(FASTEGNN_P_VIRT0_W, ld_v0, H, 1); break;
    case I_W3A: set(FASTEGNN_P_NODE0_W, ld_n0, 0, 0); break;
    case I_W3B: set(FASTEGNN_P_NODE0_W, ld_n0, H, 0); break;
    case I_W4: set(FASTEGNN_P_NODE2_W, H, 0, 0); break;
    case I_W3AT: set(FASTEGNN_P_NODE0_W, ld_n0, 0, 1); break;
    case I_W3BT: set(FASTEGNN_P_NODE0_W, ld_n0, H, 1); break;
    case I_W4T: set(FASTEGNN_P_NODE2_W, H, 0, 1); break;
    default: {
      // flat(v)[h*C + c] column of node_mlp.0.weight: W3c[o][h] = W[o][2H + h*C + c]
      int k = id - I_FIXED;
      int c = k < C ? k : k - C;
      set(FASTEGNN_P_NODE0_W, ld_n0, 2 * H + c, k < C ? 0 : 1);
      d.ks = C;
    }
  }
  return d;
}

__device__ __forceinline__ void pack_image(const PackArgs &a, int id);
__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) { pack_image(a, blockIdx.x); }
// blockIdx.y = layer
__global__ __launch_bounds__(256) void pack_all_kernel(PackArgsN an) { pack_image(an.l[blockIdx.y], blockIdx.x); }
__device__ __forceinline__ void pack_image(const PackArgs &a, int id) {
  const PackDesc d = pack_desc(a, id);
  float *dst = a.wpack + (size_t)id * IMG;
  unsigned *d3 = const_cast<unsigned *>(wpack_x3(a.wpack, a.C, id));
  auto at = [&](int o, int k) -> float {
    if (!d.src) return 0.f;
    const float w = d.transposed ? d.src[(size_t)k * d.ld + d.c0 + o * d.ks] : d.src[(size_t)o * d.ld + d.c0 + k * d.ks];
    return a.bf16 ? round_bf(w) : w;   // bf16 operand mode: every image holds the bf16-rounded weights
  };
  if (EDGE_FWD32 && !a.bf16 && (id == I_W2 || id == I_WX1)) {
    // nothing reads the fp32 images of W2 / WX1 (the edge kernels multiply on split images): their slots hold the f16x2 images of the
    // 32-edge forward kernel in the 32x32x16 operand layout (common.h, img32_word; edge_fwd32.hip).  Both plain: with the first layer
    // folded, W2's input already carries log2(e)
    unsigned *d32 = reinterpret_cast<unsigned *>(dst);
    for (int idx = threadIdx.x; idx < IMG / 2; idx += 256) {
      const int o = idx >> 5, k = (idx & 31) * 2;
      const float w0 = at(o, k), w1 = at(o, k + 1);
      d32[img32_word(0, o, k)] = split2_word(w0, w1, 0);
      d32[img32_word(1, o, k)] = split2_word(w0, w1, 1);
    }
  } else {
    for (int idx = threadIdx.x; idx < IMG; idx += 256) {
      int o = idx >> 6, k = idx & 63;
      dst[img_index(o, k)] = at(o, k);
    }
  }
  // split images: one word = two consecutive k of the same tile row (k even)
  for (int idx = threadIdx.x; idx < IMG / 2; idx += 256) {
    int o = idx >> 5, k = (idx & 31) * 2;
    const float w0 = at(o, k), w1 = at(o, k + 1);
    // log2(e) folds of the edge stage (common.h): with the FIRST layer folded, node_pre_fwd's W1a / W1b carry the factor (P and Q in
    // units of ln 2) and W2 takes an activation that is already log2(e) too large -- its image is plain; without it, W2 carries it
    if (!a.bf16 && LOG2E_FOLD_EDGE && (LOG2E_FOLD_FIRST ? (id == I_W1A || id == I_W1B) : id == I_W2)) {
      const bool f16 = img_is_f16(id, a.C);
      for (int p = 0; p < 3; ++p) d3[img3_index(p, o, k)] = split_word_d((double)w0 * LOG2E_D, (double)w1 * LOG2E_D, p, f16);
      continue;
    }
    if (!a.bf16 && img_is_f16(id, a.C)) {   // forward-only image in the f16x2 form (common.h): parts h | l, part 2 unused
      d3[img3_index(0, o, k)] = split2_word(w0, w1, 0);
      d3[img3_index(1, o, k)] = split2_word(w0, w1, 1);
      d3[img3_index(2, o, k)] = 0u;
      continue;
    }
    d3[img3_index(0, o, k)] = split_word(w0, w1, 0);
    d3[img3_index(1, o, k)] = split_word(w0, w1, 1);
    d3[img3_index(2, o, k)] = split_word(w0, w1, 2);
  }
  // row-major split image (common.h: one LDS copy serves W and W^T in the backward kernels) of V2, WXV0, WXX0, W2, WX1, W3c[c]
  int slot = rm_slot(id);
  if (id >= I_FIXED && id < I_FIXED + a.C) slot = RM_FIXED + (id - I_FIXED);
  if (slot >= 0) {
    unsigned *rm = reinterpret_cast<unsigned *>(const_cast<char *>(wpack_rm(a.wpack, a.C, slot)));
    for (int idx = threadIdx.x; idx < 64 * (RM_RS / 4); idx += 256) {
      const int o = idx / (RM_RS / 4), w = idx % (RM_RS / 4);   // word w of row o: k = 2w, 2w+1 (w >= 32: row padding)
      const float w0 = w < 32 ? at(o, 2 * w) : 0.f, w1 = w < 32 ? at(o, 2 * w + 1) : 0.f;
#pragma unroll
      for (int p = 0; p < 3; ++p) rm[p * (RM_PART / 4) + idx] = split_word(w0, w1, p);
    }
    if (slot < 5 && !a.bf16) {   // the f16x2 form of the same image (slots RM_F16 + slot; backward producers with FE_BWD_F16)
      unsigned *rf = reinterpret_cast<unsigned *>(const_cast<char *>(wpack_rm(a.wpack, a.C, RM_F16 + slot)));
      for (int idx = threadIdx.x; idx < 64 * (RM_RS / 4); idx += 256) {
        const int o = idx / (RM_RS / 4), w = idx % (RM_RS / 4);
        const float w0 = w < 32 ? at(o, 2 * w) : 0.f, w1 = w < 32 ? at(o, 2 * w + 1) : 0.f;
        rf[idx] = split2_word(w0, w1, 0);
        rf[(RM_PART / 4) + idx] = split2_word(w0, w1, 1);
        rf[2 * (RM_PART / 4) + idx] = 0u;
      }
    }
  }
}

static PackArgs pack_args(const fastegnn_layer_t *L) {
  PackArgs a;
  for (int i = 0; i < FASTEGNN_P_COUNT; ++i) a.p[i] = L->params[i];
  a.wpack = L->wpack;
  a.C = L->C;
  a.ea = L->ea;
  a.na = L->na;
  a.egnn = has(L, FASTEGNN_F_EGNN) ? 1 : 0;
  a.rf = has(L, FASTEGNN_F_RF) ? 1 : 0;
  a.bf16 = has(L, FASTEGNN_F_BF16) ? 1 : 0;
  return a;
}
int pack_weights(const fastegnn_layer_t *L, hipStream_t st) {
  FE_REQUIRE(L->params && L->wpack, "pack_weights: params/wpack null");
  if (has(L, FASTEGNN_F_WPACK_READY)) return FASTEGNN_OK;   // fastegnn_pack_weights_all has packed this layer
  const PackArgs a = pack_args(L);
  { ProfScope _ps_pack_kernel(K_PACK, st); hipLaunchKernelGGL(pack_kernel, dim3(I_FIXED + 2 * L->C), dim3(256), 0, st, a); }
  return check_launch("pack_kernel");
}
int pack_weights_all(const fastegnn_layer_t *const *layers, int n, hipStream_t st) {
  FE_REQUIRE(layers && n >= 1, "pack_weights_all: no layers");
  for (int k = 0; k < n; ++k) {
    FE_REQUIRE(layers[k] && layers[k]->params && layers[k]->wpack, "pack_weights_all: params/wpack null");
    FE_REQUIRE(layers[k]->C == layers[0]->C, "pack_weights_all: the layers of one launch share virtual_channels");
  }
  for (int k0 = 0; k0 < n; k0 += PACK_MAX_LAYERS) {
    const int m = n - k0 < PACK_MAX_LAYERS ? n - k0 : PACK_MAX_LAYERS;
    PackArgsN an;
    for (int k = 0; k < m; ++k) an.l[k] = pack_args(layers[k0 + k]);
    for (int k = m; k < PACK_MAX_LAYERS; ++k) an.l[k] = an.l[0];
    ProfScope _ps(K_PACK, st);
    hipLaunchKernelGGL(pack_all_kernel, dim3(I_FIXED + 2 * layers[0]->C, m), dim3(256), 0, st, an);
  }
  return check_launch("pack_all_kernel");
}

}  // namespace fe
