// extern "C" surface of libfastegnn_hip.so (declared in include/fastegnn_hip.h).
#include <vector>
#include "kernels.h"
#include <initializer_list>

namespace fe {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return FASTEGNN_E_LAUNCH;
  }
  return FASTEGNN_OK;
}

// ---- per-kernel HIP-event profiler -------------------------------------------------------
bool g_prof_on = false;
namespace {
struct ProfRec { int id; hipEvent_t a, b; };
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_free;
std::vector<hipEvent_t> g_open(K_COUNT, nullptr);
hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
const char *kKernelNames[K_COUNT] = {
  "pack_kernel", "node_pre_fwd_kernel", "graph_xsum_kernel", "graph_pre_fwd_kernel", "edge_fwd_kernel",
  "virt_fwd_kernel", "graph_post_fwd_kernel", "graph_post_bwd_kernel", "virt_bwd_kernel", "graph_pre_bwd_kernel",
  "edge_bwd_kernel", "edge_col_reduce_kernel", "node_pre_bwd_kernel", "wgrad_tn_kernel", "wgrad_small_kernel",
  "build_csr", "misc", "wgrad_reduce_kernel", "virt_bwd_node_kernel", "virt_bwd_gv_kernel"};
}  // namespace
void prof_begin(int id, hipStream_t st) {
  hipEvent_t e = get_event();
  (void)hipEventRecord(e, st);
  g_open[id] = e;
}
void prof_end(int id, hipStream_t st) {
  hipEvent_t e = get_event();
  (void)hipEventRecord(e, st);
  g_recs.push_back({id, g_open[id], e});
}

static int check_layer(const fastegnn_layer_t *L, const char *who) {
  if (!L) {
    set_error(std::string(who) + ": layer descriptor is null");
    return FASTEGNN_E_INVALID;
  }
  if (L->N < 0 || L->B < 1 || L->C < ((L->flags & FASTEGNN_F_EGNN) ? 0 : 1) || L->ea < 0 || L->na < 0) {
    set_error(std::string(who) + ": bad sizes");
    return FASTEGNN_E_INVALID;
  }
  if (!L->params) {
    set_error(std::string(who) + ": params null");
    return FASTEGNN_E_INVALID;
  }
  {
    const int act = (L->flags >> FASTEGNN_F_ACT_SHIFT) & FASTEGNN_F_ACT_MASK;
#ifdef FE_ACT_GENERIC
    if (act > FASTEGNN_ACT_SOFTPLUS) {
      set_error(std::string(who) + ": unknown activation kind in the flags");
      return FASTEGNN_E_INVALID;
    }
    if ((L->flags & FASTEGNN_F_BF16) && act != FASTEGNN_ACT_SILU) {
      set_error(std::string(who) + ": activations other than SiLU are not combined with the bf16 operand mode");
      return FASTEGNN_E_INVALID;
    }
#else
    if (act != FASTEGNN_ACT_SILU) {   // never a silent SiLU in place of what the caller asked for
      set_error(std::string(who) + ": this library is compiled for SiLU only; activations selected by the FASTEGNN_F_ACT bits "
                                   "need libfastegnn_hip_act.so (same sources, -DFE_ACT_GENERIC)");
      return FASTEGNN_E_INVALID;
    }
#endif
  }
  return FASTEGNN_OK;
}


// The jobs a backward stage queues into a shared batch are contracted when the batch closes: no later stage may write what
// they still have to read.  The arrays each stage writes (its outputs in fastegnn_layer_t and its region of the wg_*
// workspaces, kernels.h) are checked against the operands queued so far -- host-side pointer arithmetic, no device work
// (ADVICE round 2).  Used by fastegnn_layer_backward and by the staged entry points when L->wgrad_batch is set.
enum GuardStage { G_GRAPH_POST, G_VIRT, G_GRAPH_PRE, G_EDGE, G_NODE_PRE };
static int guard_stage(const fastegnn_layer_t *L, const WgradBatch &wb, int stage) {
  const size_t N = (size_t)L->N, BC = (size_t)L->B * L->C, rows = wg_node_rows(L), E = (size_t)L->graph.n_edges;
  const bool det = has(L, FASTEGNN_F_DETERMINISTIC);
  struct W { const float *p; size_t n; const char *what; };
  auto guard = [&](std::initializer_list<W> ws) {
    for (const W &w : ws) {
      const int r = wb.guard_write(w.p, w.n, w.what);
      if (r) return r;
    }
    return (int)FASTEGNN_OK;
  };
  switch (stage) {
    case G_VIRT:
      return guard({{L->g_h, N * H, "g_h"}, {L->g_x, N * 3, "g_x"}, {L->g_A, N * H, "g_A"}, {L->g_aggm, N * H, "g_aggm"},
                    {L->g_aggx, N * 3, "g_aggx"}, {L->g_svel, N, "g_svel"}, {L->g_sgrav, N, "g_sgrav"}, {L->g_Bc, BC * H, "g_Bc"},
                    {L->g_Zp, BC * 3, "g_Zp"}, {L->wg_node, 2 * rows * H, "wg_node[0..2)"},
                    {L->wg_virt, L->wg_virt ? fastegnn_wg_virt_floats_for(L->N, L->C, L->flags) : 0, "wg_virt"}});
    case G_GRAPH_PRE:
      return guard({{L->g_HvT, BC * H, "g_HvT"}, {L->g_Z, BC * 3, "g_Z"}, {L->g_xbar, (size_t)L->B * 4, "g_xbar"},
                    {L->wg_node ? L->wg_node + 7 * rows * H : nullptr, rows * H, "wg_node[7..8)"}});
    case G_EDGE:
      return guard({{L->g_P, N * H, "g_P"}, {L->g_xrow, N * 3, "g_xrow"}, {det ? L->g_QXe : nullptr, E * QXLD, "g_QXe"},
                    {L->g_QX_src, (size_t)L->graph.n_src * QXLD, "g_QX_src"},
                    {L->wg_edge, fastegnn_wg_edge_floats(L->graph.n_edges), "wg_edge"}});
    case G_NODE_PRE:
      return guard({{L->g_h, N * H, "g_h"}, {L->g_x, N * 3, "g_x"}, {L->g_vel, N * 3, "g_vel"},
                    {L->wg_node ? L->wg_node + 2 * rows * H : nullptr, 2 * rows * H, "wg_node[2..4)"}});
    default: return FASTEGNN_OK;
  }
}

}  // namespace fe

using namespace fe;

extern "C" int fastegnn_generic_activations(void) {
#ifdef FE_ACT_GENERIC
  return 1;
#else
  return 0;
#endif
}

#define STAGE(name, fn)                                               \
  int name(const fastegnn_layer_t *L, void *stream) {                 \
    int rc = check_layer(L, #name);                                   \
    if (rc) return rc;                                                \
    return fn(L, (hipStream_t)stream);                                \
  }

// backward stages with weight gradients: the staged entry point contracts and reduces its own jobs
#define STAGE_B(name, fn, gid)                                        \
  int name(const fastegnn_layer_t *L, void *stream) {                 \
    int rc = check_layer(L, #name);                                   \
    if (rc) return rc;                                                \
    WgradBatch *wb = static_cast<WgradBatch *>(L->wgrad_batch);       \
    if (wb && (rc = guard_stage(L, *wb, gid))) return rc;             \
    return fn(L, (hipStream_t)stream, wb);                            \
  }

extern "C" {

const char *fastegnn_last_error(void) { return g_last_error.c_str(); }
int fastegnn_version(void) { return FASTEGNN_ABI_VERSION; }
size_t fastegnn_wpack_floats(int32_t C) { return wpack_floats(C); }
size_t fastegnn_wg_slab_floats(void) { return wg_slab_floats(); }
size_t fastegnn_sizeof_layer(void) { return sizeof(fastegnn_layer_t); }
size_t fastegnn_sizeof_graph(void) { return sizeof(fastegnn_graph_t); }

STAGE(fastegnn_pack_weights, pack_weights)
STAGE(fastegnn_node_pre_forward, node_pre_forward)
STAGE(fastegnn_graph_xsum, graph_xsum)
STAGE(fastegnn_graph_pre_forward, graph_pre_forward)
STAGE(fastegnn_edge_forward, edge_forward)
STAGE(fastegnn_virt_forward, virt_forward)
STAGE(fastegnn_graph_post_forward, graph_post_forward)
STAGE_B(fastegnn_graph_post_backward, graph_post_backward, G_GRAPH_POST)
STAGE_B(fastegnn_virt_backward, virt_backward, G_VIRT)
STAGE_B(fastegnn_graph_pre_backward, graph_pre_backward, G_GRAPH_PRE)
STAGE_B(fastegnn_edge_backward, edge_backward, G_EDGE)
STAGE(fastegnn_edge_col_reduce, edge_col_reduce)
STAGE_B(fastegnn_node_pre_backward, node_pre_backward, G_NODE_PRE)

int fastegnn_pack_weights_all(const fastegnn_layer_t *const *layers, int32_t n, void *stream) {
  FE_REQUIRE(layers && n >= 1, "fastegnn_pack_weights_all: no layers");
  for (int k = 0; k < n; ++k) {
    int rc = check_layer(layers[k], "fastegnn_pack_weights_all");
    if (rc) return rc;
  }
  return pack_weights_all(layers, n, (hipStream_t)stream);
}

int fastegnn_wgrad_batch_open(const fastegnn_layer_t *L, void *stream, void **batch) {
  FE_REQUIRE(L && batch && L->wg_slab, "fastegnn_wgrad_batch_open: null argument");
  WgradBatch *wb = new WgradBatch(L->wg_slab, (hipStream_t)stream, has(L, FASTEGNN_F_BF16));
  wb->max_split = 384;   // as fastegnn_layer_backward: node-level jobs and the edge stage's slabs share 4096 slabs
  *batch = wb;
  return FASTEGNN_OK;
}
int fastegnn_wgrad_batch_close(void *batch) {
  if (!batch) return FASTEGNN_OK;
  WgradBatch *wb = static_cast<WgradBatch *>(batch);
  const int rc = wb->finish();
  delete wb;
  return rc;
}

int fastegnn_profile_enable(int32_t on) {
  g_prof_on = on != 0;
  return FASTEGNN_OK;
}
int32_t fastegnn_profile_kernels(void) { return K_COUNT; }
const char *fastegnn_profile_name(int32_t id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int fastegnn_profile_collect(double *total_ms, int64_t *launches) {
  if (!total_ms || !launches) { set_error("profile_collect: null output"); return FASTEGNN_E_INVALID; }
  for (int i = 0; i < K_COUNT; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
  for (auto &r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) {
      set_error("profile_collect: event query failed");
      return FASTEGNN_E_LAUNCH;
    }
    total_ms[r.id] += ms;
    launches[r.id] += 1;
    g_free.push_back(r.a);
    g_free.push_back(r.b);
  }
  g_recs.clear();
  return FASTEGNN_OK;
}

int fastegnn_layer_forward(const fastegnn_layer_t *L, void *stream) {
  int rc = check_layer(L, "fastegnn_layer_forward");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = pack_weights(L, st))) return rc;
  if ((rc = node_pre_forward(L, st))) return rc;
  if (has(L, FASTEGNN_F_EGNN)) {   // EGNN baseline: no virtual nodes, hence no per-graph stages
    if ((rc = edge_forward(L, st))) return rc;
    return virt_forward(L, st);
  }
  if ((rc = graph_xsum(L, st))) return rc;
  if ((rc = graph_pre_forward(L, st))) return rc;
  if ((rc = edge_forward(L, st))) return rc;
  if ((rc = virt_forward(L, st))) return rc;
  return graph_post_forward(L, st);
}

int fastegnn_layer_backward(const fastegnn_layer_t *L, void *stream) {
  int rc = check_layer(L, "fastegnn_layer_backward");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  // the node-level / graph-level weight-gradient jobs of all stages and the edge stage's slabs are contracted and reduced
  // together at the end of the layer: one wgrad_tn launch and one wgrad_reduce launch instead of a pair per stage
  FE_REQUIRE(L->wg_slab, "fastegnn_layer_backward: wg_slab null");
  WgradBatch wb(L->wg_slab, st);
  // up to eight node-level jobs of N rows + the edge stage's 2 x 256 slabs share the lower half of the slab workspace
  // (4096 slabs): 384 partial slabs per job at most (at cfg4 sizes a job takes 256 anyway)
  wb.max_split = 384;
  auto guard_virt = [&]() { return guard_stage(L, wb, G_VIRT); };
  auto guard_edge = [&]() { return guard_stage(L, wb, G_EDGE); };
  auto guard_node_pre = [&]() { return guard_stage(L, wb, G_NODE_PRE); };
  if (has(L, FASTEGNN_F_EGNN)) {
    if ((rc = guard_virt())) return rc;
    if ((rc = virt_backward(L, st, &wb))) return rc;
    if ((rc = guard_edge())) return rc;
    if ((rc = edge_backward(L, st, &wb))) return rc;
    if ((rc = edge_col_reduce(L, st))) return rc;
    if ((rc = guard_node_pre())) return rc;
    if ((rc = node_pre_backward(L, st, &wb))) return rc;
    return wb.finish();
  }
  if ((rc = graph_post_backward(L, st, &wb))) return rc;
  if ((rc = guard_virt())) return rc;
  if ((rc = virt_backward(L, st, &wb))) return rc;
  if ((rc = guard_stage(L, wb, G_GRAPH_PRE))) return rc;
  if ((rc = graph_pre_backward(L, st, &wb))) return rc;
  if ((rc = guard_edge())) return rc;
  if ((rc = edge_backward(L, st, &wb))) return rc;
  if ((rc = edge_col_reduce(L, st))) return rc;
  if ((rc = guard_node_pre())) return rc;
  if ((rc = node_pre_backward(L, st, &wb))) return rc;
  return wb.finish();
}

}  // extern "C"
