// Internal launcher interface between api.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/fastegnn_hip.h"
#include "common.h"

namespace fe {

void set_error(const std::string &msg);
int check_launch(const char *what);

#define FE_REQUIRE(cond, msg)            \
  do {                                   \
    if (!(cond)) {                       \
      fe::set_error(msg);                \
      return FASTEGNN_E_INVALID;         \
    }                                    \
  } while (0)

inline bool has(const fastegnn_layer_t *L, int f) { return (L->flags & f) != 0; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- per-kernel HIP-event profiler (api.hip); off unless fastegnn_profile_enable(1) ----
enum KernelId {
  K_PACK = 0, K_NODE_PRE_FWD, K_XSUM, K_GRAPH_PRE_FWD, K_EDGE_FWD, K_VIRT_FWD, K_GRAPH_POST_FWD,
  K_GRAPH_POST_BWD, K_VIRT_BWD, K_GRAPH_PRE_BWD, K_EDGE_BWD, K_COL_REDUCE, K_NODE_PRE_BWD,
  K_WGRAD_TN, K_WGRAD_SMALL, K_CSR, K_MISC, K_WGRAD_REDUCE, K_VIRT_BWD_NODE, K_VIRT_BWD_GV, K_COUNT
};
extern bool g_prof_on;
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);
struct ProfScope {
  int id; hipStream_t st; bool on;
  ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), on(g_prof_on) { if (on) prof_begin(id, st); }
  ~ProfScope() { if (on) prof_end(id, st); }
};

// generic weight-gradient contraction (misc.hip):
//   dW[o*lddw + c0 + k*ks] += sum_m G[m*ldg + o] * T[m*ldt + k]   (o < 64, k < kmax),  db[o] += sum_m G[m*ldg+o]
//   batched over `nb` with strides (sG, sT, sW) in floats.  Jobs are queued and run by finish().
constexpr int WGV_PAD = 16;        // padding rows (x C) behind each [N*C,64] operand array of the virtual backward
constexpr int WG_MAX_JOBS = 24;
constexpr int WG_SLABS = 8192;   // 64x64 partial slabs in the wg_slab workspace (+ 64-float bias slabs); the upper half is the
                                 // virtual stage's own batch (WgradBatch slab_base)
struct WgJob {
  const float *G, *T;
  float *dW, *db;
  long M, sG, sT, sW;
  int ldg, ldt, lddw, c0, ks, kmax, rows_per_wg, nsplit, nb, wg_begin, slab_begin;
  int round;   // bf16 operand mode: both operands are rounded to bf16 before the product
};
struct WgTable {
  WgJob job[WG_MAX_JOBS];
  float *slab, *slab_b;
  int n_jobs;
};
struct WgradBatch {
  WgTable tab;
  hipStream_t st;
  int n_wg, n_slab, max_nb;
  bool round;   // applied to the jobs added from now on
  int min_rows; // smallest row range given to one workgroup (short operands are split that far to fill the chip)
  int max_split; // most workgroups (= partial slabs) one job may take, times its batch count
  int slab_base, slab_cap;   // this batch's share of the slab workspace: [slab_base, slab_base + slab_cap)
  int slab_top;              // slab jobs (add_slabs) take their ranges downwards from the top of the share
  bool planned;              // the contraction jobs have their row / workgroup / slab ranges
  WgradBatch(float *slab, hipStream_t st, bool round_bf16 = false, int slab_base = 0, int slab_cap = WG_SLABS / 2);
  int add(const float *G, int ldg, const float *T, int ldt, long M, float *dW, int lddw, int c0, int ks, float *db,
          int nb = 1, long sG = 0, long sT = 0, long sW = 0, int kmax = 64);
  // a job whose partial slabs (nsplit of them, [64][64] + [64] bias each) are written by the caller's own kernel:
  // only the fixed-order reduction into dW / db runs here.  *slab_begin receives the first slab index.
  int plan();
  int add_slabs(float *dW, int lddw, int c0, int ks, float *db, int nsplit, int *slab_begin);
  // the same for `nb` weights whose partial slabs the caller keeps in an array of its OWN: ext[(b * nsplit + split) * 4096], no bias,
  // elements in the accumulator order of the 32x32 MFMA blocks (common.h, WgAcc32); weight b adds into dW + b * sW
  int add_slabs_ext(const float *ext, float *dW, int lddw, int c0, int ks, int nsplit, int nb, long sW);
  int finish();
  // The contractions are deferred to finish(): a job's operand rows must stay untouched until then.  A stage that is about
  // to write `n` floats at `p` while the batch is open declares it here; an overlap with an operand of a queued job is an
  // error (host-side check, no device work).
  int guard_write(const float *p, size_t n, const char *what) const;
};
inline size_t wg_slab_floats() { return (size_t)WG_SLABS * (IMG + H); }
//   dW[o*lddw + c0 + a] += sum_m G[m*ldg + o] * F[m*ldf + a],  a < kf <= 8
int launch_dgrad_small(const float *G, long N, int kf, const float *W, int ldw, int c0, float *g_in, int accumulate,
                       hipStream_t st);
int launch_wgrad_small(const float *G, int ldg, const float *F, int ldf, int kf, long M, float *dW, int lddw, int c0,
                       hipStream_t st);

// stage launchers
int pack_weights(const fastegnn_layer_t *L, hipStream_t st);
int pack_weights_all(const fastegnn_layer_t *const *layers, int n, hipStream_t st);
int node_pre_forward(const fastegnn_layer_t *L, hipStream_t st);
int graph_xsum(const fastegnn_layer_t *L, hipStream_t st);
int graph_pre_forward(const fastegnn_layer_t *L, hipStream_t st);
int edge_forward(const fastegnn_layer_t *L, hipStream_t st);
// edge_fwd32.hip: the same stage on 32-edge tiles / 32x32x16 MFMAs (default build, fp32-grade SiLU mode)
bool edge_forward32_applies(const fastegnn_layer_t *L);
int edge_forward32(const fastegnn_layer_t *L, hipStream_t st);
int virt_forward(const fastegnn_layer_t *L, hipStream_t st);
int graph_post_forward(const fastegnn_layer_t *L, hipStream_t st);
// `shared`: a weight-gradient batch that outlives the stage (fastegnn_layer_backward: ONE contraction launch and ONE
// reduction launch per layer instead of one pair per stage); null: the stage contracts and reduces its own jobs.
int graph_post_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared = nullptr);
int virt_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared = nullptr);
int graph_pre_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared = nullptr);
int edge_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared = nullptr);
// virt_bwd.hip: floats of wg_virt for B4 (1 <= C <= 64)
size_t virt_pc_wg_floats(size_t N, size_t C);
// the channel-phased form of B4 (virt_bwd_cs_kernel): does it run for this shape, and its share of wg_virt
bool virt_cs_applies(long N, int C, int flags);
size_t virt_cs_wg_floats(size_t C);
int edge_col_reduce(const fastegnn_layer_t *L, hipStream_t st);
int node_pre_backward(const fastegnn_layer_t *L, hipStream_t st, WgradBatch *shared = nullptr);
// operand regions of the node-level / graph-level weight gradients inside wg_node (disjoint, so that the jobs of a whole
// layer can be contracted together): [0,2M) virt, [2M,4M) node_pre, [4M,7M) graph_post, [7M,8M) graph_pre rows of 64 floats,
// M = max(N, B*C)
inline size_t wg_node_rows(const fastegnn_layer_t *L) { const size_t bc = (size_t)L->B * L->C; return (size_t)L->N > bc ? (size_t)L->N : bc; }

}  // namespace fe
