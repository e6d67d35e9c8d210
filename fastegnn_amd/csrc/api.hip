// extern "C" surface of libfastegnn_hip.so (declared in include/fastegnn_hip.h).
#include "kernels.h"

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

static int check_layer(const fastegnn_layer_t *L, const char *who) {
  if (!L) {
    set_error(std::string(who) + ": layer descriptor is null");
    return FASTEGNN_E_INVALID;
  }
  if (L->N < 0 || L->B < 1 || L->C < 1 || L->ea < 0 || L->na < 0) {
    set_error(std::string(who) + ": bad sizes");
    return FASTEGNN_E_INVALID;
  }
  if (!L->params) {
    set_error(std::string(who) + ": params null");
    return FASTEGNN_E_INVALID;
  }
  return FASTEGNN_OK;
}

}  // namespace fe

using namespace fe;

#define STAGE(name, fn)                                               \
  int name(const fastegnn_layer_t *L, void *stream) {                 \
    int rc = check_layer(L, #name);                                   \
    if (rc) return rc;                                                \
    return fn(L, (hipStream_t)stream);                                \
  }

extern "C" {

const char *fastegnn_last_error(void) { return g_last_error.c_str(); }
int fastegnn_version(void) { return 100; }
size_t fastegnn_wpack_floats(int32_t C) { return wpack_floats(C); }

STAGE(fastegnn_pack_weights, pack_weights)
STAGE(fastegnn_node_pre_forward, node_pre_forward)
STAGE(fastegnn_graph_xsum, graph_xsum)
STAGE(fastegnn_graph_pre_forward, graph_pre_forward)
STAGE(fastegnn_edge_forward, edge_forward)
STAGE(fastegnn_virt_forward, virt_forward)
STAGE(fastegnn_graph_post_forward, graph_post_forward)
STAGE(fastegnn_graph_post_backward, graph_post_backward)
STAGE(fastegnn_virt_backward, virt_backward)
STAGE(fastegnn_graph_pre_backward, graph_pre_backward)
STAGE(fastegnn_edge_backward, edge_backward)
STAGE(fastegnn_edge_col_reduce, edge_col_reduce)
STAGE(fastegnn_node_pre_backward, node_pre_backward)

int fastegnn_layer_forward(const fastegnn_layer_t *L, void *stream) {
  int rc = check_layer(L, "fastegnn_layer_forward");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = pack_weights(L, st))) return rc;
  if ((rc = node_pre_forward(L, st))) return rc;
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
  if ((rc = graph_post_backward(L, st))) return rc;
  if ((rc = virt_backward(L, st))) return rc;
  if ((rc = graph_pre_backward(L, st))) return rc;
  if ((rc = edge_backward(L, st))) return rc;
  if ((rc = edge_col_reduce(L, st))) return rc;
  return node_pre_backward(L, st);
}

}  // extern "C"
