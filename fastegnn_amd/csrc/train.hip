// Training-step closure on device: the caller-side pieces of the reference
// harness that bracket the hot path in utils/train.py -- edge_attr augmentation (:41-43), MSE + MMD loss
// with its gradient (:104-165, kernel() :17-20) and Adam (main_nbody.py:137) -- so that one training
// iteration never leaves the GPU.
#include "kernels.h"

namespace fe {

// out[e] = [edge_attr[e,:], ||loc[row_e] - loc[col_e]||]        (utils/train.py:41-43)
__global__ void augment_edge_attr_kernel(const int64_t *ei, const float *loc, const float *ea, int E, int k, float *out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const long r = ei[e], c = ei[(size_t)E + e];
  float d0 = loc[r * 3] - loc[c * 3], d1 = loc[r * 3 + 1] - loc[c * 3 + 1], d2 = loc[r * 3 + 2] - loc[c * 3 + 2];
  for (int a = 0; a < k; ++a) out[(size_t)e * (k + 1) + a] = ea[(size_t)e * k + a];
  out[(size_t)e * (k + 1) + k] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
}

// MSE part: loss[1] (and loss[0]) += sum (p-t)^2 / (3N);  g_loc = 2 (p-t) / (3N)
__global__ __launch_bounds__(256) void loss_mse_kernel(const float *pred, const float *tgt, long n3, float *g_loc, float *loss) {
  __shared__ float red[4];
  const float inv = 1.0f / (float)n3;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n3; i += (long)gridDim.x * 256) {
    const float d = pred[i] - tgt[i];
    s += d * d;
    g_loc[i] = 2.f * d * inv;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1] + red[2] + red[3]) * inv;
    atomicAdd(&loss[0], v);
    atomicAdd(&loss[1], v);
  }
}

// MMD part, one workgroup per graph: l_vv = sum_{c,c'} k(V_c,V_c') / (B C^2), l_rv = 2 sum_{s,c} k(R_s,V_c) / (B S C),
// k(x,y) = exp(-||x-y|| / (2 sigma^2));  loss[0] += weight (l_vv - l_rv);  gradients into g_vloc [B,3,C] and g_loc.
__global__ __launch_bounds__(256) void loss_mmd_kernel(const float *pred, const float *vloc, const int32_t *samp, int B,
                                                       int C, int S, float sigma, float weight, float *g_loc,
                                                       float *g_vloc, float *loss) {
  extern __shared__ float sm[];
  float *V = sm;             // [C][3]
  float *gV = sm + 3 * C;    // [C][3]
  float *R = gV + 3 * C;     // [S][3]
  float *gR = R + 3 * S;     // [S][3]
  __shared__ float acc;
  const int b = blockIdx.x;
  const float i2s = 1.0f / (2.f * sigma * sigma);
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    int c = i / 3, k = i % 3;
    V[i] = vloc[((size_t)b * 3 + k) * C + c];
    gV[i] = 0.f;
  }
  for (int i = threadIdx.x; i < 3 * S; i += 256) {
    int s = i / 3, k = i % 3;
    R[i] = pred[(size_t)samp[b * S + s] * 3 + k];
    gR[i] = 0.f;
  }
  if (threadIdx.x == 0) acc = 0.f;
  __syncthreads();
  const float w_vv = weight / ((float)B * C * C), w_rv = -2.f * weight / ((float)B * S * C);
  float part = 0.f;
  for (int i = threadIdx.x; i < C * C + S * C; i += 256) {
    const bool vv = i < C * C;
    const int a = vv ? i / C : (i - C * C) / C, c = vv ? i % C : (i - C * C) % C;
    const float *xa = vv ? V + 3 * a : R + 3 * a;
    float *ga = vv ? gV + 3 * a : gR + 3 * a;
    const float d0 = xa[0] - V[3 * c], d1 = xa[1] - V[3 * c + 1], d2 = xa[2] - V[3 * c + 2];
    const float dist = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float kv = __expf(-dist * i2s);
    const float w = vv ? w_vv : w_rv;
    part += w * kv;
    if (dist > 0.f) {   // d/dx exp(-|x-y|/2s^2) = -k/(2s^2) (x-y)/|x-y|; zero at coincident points (cdist backward)
      const float f = -w * kv * i2s / dist;
      atomicAdd(&ga[0], f * d0); atomicAdd(&ga[1], f * d1); atomicAdd(&ga[2], f * d2);
      atomicAdd(&gV[3 * c], -f * d0); atomicAdd(&gV[3 * c + 1], -f * d1); atomicAdd(&gV[3 * c + 2], -f * d2);
    }
  }
  atomicAdd(&acc, part);
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&loss[0], acc);
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    int c = i / 3, k = i % 3;
    g_vloc[((size_t)b * 3 + k) * C + c] = gV[i];
  }
  for (int i = threadIdx.x; i < 3 * S; i += 256) {
    int s = i / 3, k = i % 3;
    atomicAdd(&g_loc[(size_t)samp[b * S + s] * 3 + k], gR[i]);
  }
}

// torch.optim.Adam (no amsgrad; weight decay folded into the gradient), up to 24 tensors per launch
constexpr int ADAM_MAX = 24;
struct AdamArgs {
  float *p[ADAM_MAX], *m[ADAM_MAX], *v[ADAM_MAX];
  const float *g[ADAM_MAX];
  long n[ADAM_MAX];
  int count;
  float lr_t, b1, b2, inv_bc2_sqrt, eps, wd;
};
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  const int t = blockIdx.y;
  if (t >= a.count) return;
  float *p = a.p[t], *m = a.m[t], *v = a.v[t];
  const float *g = a.g[t];
  if (!g) return;   // torch.optim.Adam skips parameters whose .grad is None: no decay, no moment update
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n[t]; i += (long)gridDim.x * 256) {
    const float gi = g[i] + a.wd * p[i];
    const float mi = a.b1 * m[i] + (1.f - a.b1) * gi;
    const float vi = a.b2 * v[i] + (1.f - a.b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= a.lr_t * mi / (sqrtf(vi) * a.inv_bc2_sqrt + a.eps);
  }
}

}  // namespace fe

using namespace fe;

extern "C" {

int fastegnn_augment_edge_attr(const int64_t *edge_index, const float *loc, const float *edge_attr, int32_t E,
                               int32_t k, float *out, void *stream) {
  if (E == 0) return FASTEGNN_OK;
  FE_REQUIRE(edge_index && loc && out && (k == 0 || edge_attr), "augment_edge_attr: null pointer");
  hipLaunchKernelGGL(augment_edge_attr_kernel, dim3(cdiv(E, 256)), dim3(256), 0, (hipStream_t)stream, edge_index, loc,
                     edge_attr, E, k, out);
  return check_launch("augment_edge_attr_kernel");
}

int fastegnn_loss_mse_mmd(const float *loc_pred, const float *loc_t, const float *vloc, const int32_t *sample_nodes,
                          int32_t N, int32_t B, int32_t C, int32_t S, float sigma, float weight, float *loss2,
                          float *g_loc, float *g_vloc, void *stream) {
  FE_REQUIRE(loc_pred && loc_t && vloc && loss2 && g_loc && g_vloc && (S == 0 || sample_nodes),
             "loss_mse_mmd: null pointer");
  FE_REQUIRE(C <= 256 && S <= 4096, "loss_mse_mmd: C <= 256 and S <= 4096");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(loss2, 0, 2 * sizeof(float), st);
  int grid = cdiv((long)N * 3, 256 * 8);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(loss_mse_kernel, dim3(grid), dim3(256), 0, st, loc_pred, loc_t, (long)N * 3, g_loc, loss2);
  const size_t lds = (size_t)(6 * C + 6 * S) * sizeof(float);
  hipLaunchKernelGGL(loss_mmd_kernel, dim3(B), dim3(256), lds, st, loc_pred, vloc, sample_nodes, B, C, S, sigma, weight,
                     g_loc, g_vloc, loss2);
  return check_launch("loss_mse_mmd");
}

int fastegnn_adam_step(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                       const int64_t *numel, int32_t n_tensors, int32_t step, float lr, float beta1, float beta2,
                       float eps, float weight_decay, void *stream) {
  FE_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && step >= 1, "adam_step: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  for (int base = 0; base < n_tensors; base += ADAM_MAX) {
    AdamArgs a;
    a.count = n_tensors - base < ADAM_MAX ? n_tensors - base : ADAM_MAX;
    long nmax = 0;
    for (int t = 0; t < a.count; ++t) {
      a.p[t] = params[base + t]; a.g[t] = grads[base + t]; a.m[t] = exp_avg[base + t]; a.v[t] = exp_avg_sq[base + t];
      a.n[t] = numel[base + t];
      if (a.n[t] > nmax) nmax = a.n[t];
    }
    a.lr_t = (float)(lr / bc1); a.b1 = beta1; a.b2 = beta2; a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)); a.eps = eps;
    a.wd = weight_decay;
    int gx = cdiv(nmax, 256 * 4);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(adam_kernel, dim3(gx, a.count), dim3(256), 0, st, a);
  }
  return check_launch("adam_kernel");
}

}  // extern "C"
