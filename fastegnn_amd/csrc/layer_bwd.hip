#include "kernels.h"
namespace fe {
#define NOTIMPL(name) int name(const fastegnn_layer_t *, hipStream_t) { set_error(#name ": not implemented"); return FASTEGNN_E_INVALID; }
NOTIMPL(graph_post_backward)
NOTIMPL(virt_backward)
NOTIMPL(graph_pre_backward)
NOTIMPL(edge_backward)
NOTIMPL(edge_col_reduce)
NOTIMPL(node_pre_backward)
}
