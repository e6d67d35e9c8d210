"""ctypes binding of libfastegnn_hip.so (C ABI: include/fastegnn_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C fastegnn_amd/csrc``.
There is no CPU fallback: if the shared object is missing, or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FASTEGNN_SAFE_WAITS=1 loads the -DFE_SAFE_WAITS build of the same sources (csrc/Makefile, `make safe`): every hand-counted wait,
# LDS-DMA copy and relaxed LDS flag in its conservative form.  Same ABI, same results; a diagnostic, not a fallback.
SAFE_WAITS = os.environ.get("FASTEGNN_SAFE_WAITS", "0") not in ("", "0")
# Operand range.  The default build multiplies on 2-part fp16 splits (f16x2): hidden activations and [64,64] weights must stay below
# 65 504 in magnitude, which the reference's plain fp32 (models/FastEGNN.py:102-119) does not require.  The modules therefore guard
# every forward (fastegnn_check_finite on the outputs, into a host-mapped word that is polled without synchronisation) and move to the
# wide-range build (libfastegnn_hip_x3.so / _act_x3.so: 3-part bf16 splits, fp32's exponent range, ~8 % slower) once a pass has left
# the range, staying there (fastegnn_amd.model.RangeGuard; FASTEGNN_RANGE_CHECK=sync re-runs the overflowing call itself).
#   FASTEGNN_WIDE_RANGE unset : automatic (above)        =1 : wide-range build from the first call        =0 : f16x2 build, overflow raises
_wr = os.environ.get("FASTEGNN_WIDE_RANGE", "")
WIDE_RANGE = None if _wr == "" else _wr != "0"
if SAFE_WAITS and WIDE_RANGE:
    raise RuntimeError("fastegnn_amd: FASTEGNN_SAFE_WAITS=1 (a diagnostic build of the f16x2 arithmetic) cannot be combined with "
                       "FASTEGNN_WIDE_RANGE=1")


def lib_path(act: bool = False, wide: bool = False) -> str:
    if SAFE_WAITS:
        if act:
            raise RuntimeError("fastegnn_amd: FASTEGNN_SAFE_WAITS=1 has no generic-activation build (act_fn must be SiLU)")
        return os.path.join(_HERE, "libfastegnn_hip_safe.so")
    return os.path.join(_HERE, "libfastegnn_hip" + ("_act" if act else "") + ("_x3" if wide else "") + ".so")


LIB_PATH = lib_path(False, bool(WIDE_RANGE))   # the library the stage-independent helpers (CSR, graphs, training step, comm) use

ABI_VERSION = 107   # FASTEGNN_ABI_VERSION of include/fastegnn_hip.h this mirror was written against
H = 64
QX_LD = 68
FEATW = 8

# flags (fastegnn_hip.h)
F_ATTENTION, F_NORMALIZE, F_TANH, F_RESIDUAL, F_GRAVITY, F_COORDS_SUM, F_EGNN, F_RF, F_BF16, F_EGNN_NORM = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
F_WPACK_READY = 65536   # the layer's weight images were packed by fastegnn_pack_weights_all
F_GQX_ACCUM = 32768   # edge_backward scatters into g_QX_src without zeroing it (second launch of a layer; include/fastegnn_hip.h)
F_DETERMINISTIC = 1024   # backward: per-edge rows + CSC reduce instead of the atomic scatter (include/fastegnn_hip.h)


def deterministic_default():
    """FASTEGNN_DETERMINISTIC=1 / =0 selects the order-independent col-side sums (store + CSC reduce) / the atomic scatter for every model that does
    not say otherwise; unset (None): each model picks the faster one for its operand mode (FastEGNN.deterministic)."""
    v = os.environ.get("FASTEGNN_DETERMINISTIC")
    return None if v is None or v == "" else v == "1"

# per-layer parameter slots, in header order -> reference state_dict suffix (models/FastEGNN.py:28-99)
PARAM_SLOTS = [
    "edge_mlp.0.weight", "edge_mlp.0.bias", "edge_mlp.2.weight", "edge_mlp.2.bias",
    "edge_mlp_virtual.0.weight", "edge_mlp_virtual.0.bias", "edge_mlp_virtual.2.weight", "edge_mlp_virtual.2.bias",
    "att_mlp.0.weight", "att_mlp.0.bias", "att_mlp_virtual.0.weight", "att_mlp_virtual.0.bias",
    "coord_mlp_r.0.weight", "coord_mlp_r.0.bias", "coord_mlp_r.2.weight",
    "coord_mlp_r_virtual.0.weight", "coord_mlp_r_virtual.0.bias", "coord_mlp_r_virtual.2.weight",
    "coord_mlp_v_virtual.0.weight", "coord_mlp_v_virtual.0.bias", "coord_mlp_v_virtual.2.weight",
    "coord_mlp_vel.0.weight", "coord_mlp_vel.0.bias", "coord_mlp_vel.2.weight", "coord_mlp_vel.2.bias",
    "gravity_mlp.0.weight", "gravity_mlp.0.bias", "gravity_mlp.2.weight", "gravity_mlp.2.bias",
    "node_mlp.0.weight", "node_mlp.0.bias", "node_mlp.2.weight", "node_mlp.2.bias",
    "node_mlp_virtual.0.weight", "node_mlp_virtual.0.bias", "node_mlp_virtual.2.weight", "node_mlp_virtual.2.bias",
    "coord_mlp_r.2.bias",     # EGNN baseline only (FastEGNN's coordinate heads have no bias)
]
P_COUNT = len(PARAM_SLOTS)
assert P_COUNT == 38

_vp = C.c_void_p
_i32 = C.c_int32


class GraphT(C.Structure):
    _fields_ = [("n_rows", _i32), ("n_src", _i32), ("n_edges", _i32), ("n_chunks", _i32),
                ("rowptr", _vp), ("erow", _vp), ("col", _vp), ("perm", _vp),
                ("cscptr", _vp), ("csc_eid", _vp), ("chunk_row", _vp)]


class PadDesc(C.Structure):
    """fastegnn_pad_desc_t (include/fastegnn_hip.h)"""
    _fields_ = [("src", _vp), ("dst", _vp), ("rows", _i32), ("cols", _i32), ("rows_dst", _i32), ("cols_dst", _i32),
                ("nblk", _i32), ("blk", _i32 * 3), ("lead", _i32), ("reserved", _i32)]


_LAYER_PTRS_A = ["batch", "gptr", "ea_sorted", "vel", "node_attr", "params", "grads", "wpack",
                 "h", "x", "Z", "HvT", "h_out", "x_out", "Z_out", "HvT_out",
                 "P", "QX", "QX_src", "A", "svel", "sgrav", "xsum", "Bc", "aggm", "aggx", "npre", "poolV", "poolX",
                 "g_h_out", "g_x_out", "g_Z_out", "g_HvT_out", "g_h", "g_x", "g_Z", "g_HvT", "g_vel", "g_ea_sorted", "g_node_attr",
                 "g_poolV", "g_poolX", "g_Bc", "g_Zp", "g_xbar", "g_A", "g_P", "g_aggm", "g_aggx",
                 "g_svel", "g_sgrav", "g_QXe", "g_QX_src", "g_QX", "g_xrow", "wg_edge", "wg_virt", "wg_node", "wg_slab", "wgrad_batch"]


class LayerT(C.Structure):
    _fields_ = ([("N", _i32), ("B", _i32), ("C", _i32), ("ea", _i32), ("na", _i32), ("flags", _i32),
                 ("gravity", C.c_float * 3), ("epsilon", C.c_float), ("act_param", C.c_float),
                 ("graph", GraphT)]
                + [(n, _vp) for n in _LAYER_PTRS_A])


_libs = {}
ACT_LIB_PATH = lib_path(True, bool(WIDE_RANGE)) if not SAFE_WAITS else None
# activation kinds of the FASTEGNN_F_ACT bits (include/fastegnn_hip.h)
F_ACT_SHIFT = 11
ACT_SILU, ACT_RELU, ACT_LEAKY_RELU, ACT_TANH, ACT_SIGMOID, ACT_ELU, ACT_GELU, ACT_SOFTPLUS = range(8)
ACT_NONE = -1   # fastegnn_wide_linear / _dx / _dw: no fused activation


def lib(act: bool = False, wide=None):
    """Load a build of the HIP library; raises (never falls back to a CPU path) when it is missing.
    act=True: the generic-activation build (-DFE_ACT_GENERIC) -- the layer / stage calls of a model whose act_fn is not SiLU go
    there; the default library rejects their flags.  wide: True = the wide-range (bf16x3) form, False = the f16x2 form,
    None = what FASTEGNN_WIDE_RANGE selects (f16x2 unless it is 1)."""
    path = lib_path(act, bool(WIDE_RANGE) if wide is None else bool(wide))
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"fastegnn_amd: {path} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fastegnn_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(path)
    if bool(L.fastegnn_generic_activations()) != act:
        raise RuntimeError(f"fastegnn_amd: {path} is not the {'generic-activation' if act else 'SiLU'} build")
    if bool(L.fastegnn_f16_operands()) == path.endswith("_x3.so"):
        raise RuntimeError(f"fastegnn_amd: {path} was not built with the arithmetic its name says (csrc/Makefile)")
    L.fastegnn_last_error.restype = C.c_char_p
    L.fastegnn_version.restype = C.c_int
    if L.fastegnn_version() != ABI_VERSION:
        raise RuntimeError(f"fastegnn_amd: {path} has ABI revision {L.fastegnn_version()}, this binding expects "
                           f"{ABI_VERSION} -- rebuild with `make -C fastegnn_amd/csrc`")
    L.fastegnn_wpack_floats.restype = C.c_size_t
    L.fastegnn_wpack_floats.argtypes = [_i32]
    L.fastegnn_csr_tmp_bytes.restype = C.c_size_t
    L.fastegnn_csr_tmp_bytes.argtypes = [_i32, _i32, _i32]
    L.fastegnn_chunk_rows.restype = C.c_size_t
    L.fastegnn_chunk_rows.argtypes = [_i32]
    L.fastegnn_wg_edge_floats.restype = C.c_size_t
    L.fastegnn_wg_edge_floats.argtypes = [_i32]
    L.fastegnn_wg_virt_floats.restype = C.c_size_t
    L.fastegnn_wg_virt_floats.argtypes = [_i32, _i32]
    L.fastegnn_wg_virt_floats_for.restype = C.c_size_t
    L.fastegnn_wg_virt_floats_for.argtypes = [_i32, _i32, _i32]
    L.fastegnn_backward_scratch_floats_for.restype = C.c_size_t
    L.fastegnn_backward_scratch_floats_for.argtypes = [_i32, _i32, _i32, _i32, _i32, _i32]
    L.fastegnn_wg_node_floats.restype = C.c_size_t
    L.fastegnn_wg_node_floats.argtypes = [_i32, _i32, _i32]
    L.fastegnn_backward_scratch_floats.restype = C.c_size_t
    L.fastegnn_backward_scratch_floats.argtypes = [_i32, _i32, _i32, _i32, _i32]
    L.fastegnn_chunk_edges.restype = _i32
    L.fastegnn_chunk_edges.argtypes = []
    L.fastegnn_pad_params.argtypes = [C.POINTER(PadDesc), _i32, _i32, _i32, _vp]
    L.fastegnn_build_csr.argtypes = [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     C.POINTER(_i32), _vp, C.c_size_t, _vp]
    L.fastegnn_permute_rows.argtypes = [_vp, _vp, _i32, _i32, _vp, _vp]
    L.fastegnn_build_batch.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp]
    L.fastegnn_embed_forward.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _vp]
    L.fastegnn_embed_backward.argtypes = [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]
    L.fastegnn_virtual_init.argtypes = [_vp, _i32, _i32, _vp, _vp]
    L.fastegnn_virtual_init_backward.argtypes = [_vp, _i32, _i32, _vp, _vp]
    L.fastegnn_augment_edge_attr.argtypes = [_vp, _vp, _vp, _i32, _i32, _vp, _vp]
    L.fastegnn_loss_mse_mmd.argtypes = [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.c_float, C.c_float, _vp, _vp, _vp, _vp]
    L.fastegnn_adam_step.argtypes = [_vp, _vp, _vp, _vp, C.POINTER(C.c_int64), _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, _vp]
    L.fastegnn_radius_graph_ws_bytes.restype = C.c_size_t
    L.fastegnn_radius_graph_ws_bytes.argtypes = [_i32]
    L.fastegnn_radius_graph_count.argtypes = [_vp, _i32, C.c_float, _vp, C.c_size_t, C.POINTER(C.c_int64), _vp]
    L.fastegnn_radius_graph_fill.argtypes = [_vp, _i32, C.c_float, _vp, C.c_size_t, C.c_int64, _vp, _vp, _vp]
    L.fastegnn_cutoff_tmp_bytes.restype = C.c_size_t
    L.fastegnn_cutoff_tmp_bytes.argtypes = [C.c_int64]
    L.fastegnn_cutoff_edges.argtypes = [_vp, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_size_t, _vp]
    L.fastegnn_nbody_cutoff_edges.argtypes = [_vp, _i32, _i32, _i32, _vp, _vp, _vp]
    L.fastegnn_selftest_gemm.argtypes = [_vp, _vp, _vp, _i32, _vp]
    L.fastegnn_selftest_rm.argtypes = [_vp, _vp, _vp, _i32, _i32, _vp]
    L.fastegnn_selftest_jreduce.argtypes = [_vp, _vp, _vp]
    L.fastegnn_selftest_lane_sums.argtypes = [_vp, _vp, _vp]
    L.fastegnn_selftest_chain.argtypes = [_vp, _vp, _i32, _i32, _i32, _i32, _vp]
    L.fastegnn_selftest_chain_bf3.argtypes = [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]
    L.fastegnn_selftest_wgrad.argtypes = [_vp, _vp, _i32, _vp, _vp, _vp, _vp]
    L.fastegnn_selftest_stream.argtypes = [_vp, _vp, C.c_size_t, _i32, _vp]
    L.fastegnn_selftest_wgrad_guard.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.fastegnn_selftest_wgrad_plan.argtypes = [C.POINTER(C.c_int64), C.POINTER(_i32), _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]
    L.fastegnn_comm_unique_id_bytes.restype = _i32
    L.fastegnn_comm_unique_id.argtypes = [_vp]
    L.fastegnn_comm_init.argtypes = [C.POINTER(_vp), _vp, _i32, _i32]
    L.fastegnn_comm_destroy.argtypes = [_vp]
    L.fastegnn_comm_rank.argtypes = [_vp]
    L.fastegnn_comm_world.argtypes = [_vp]
    L.fastegnn_comm_all_reduce.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.fastegnn_comm_all_gather.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp]
    L.fastegnn_comm_reduce_scatter.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp]
    L.fastegnn_comm_all_to_all_v.argtypes = [_vp, _vp, C.POINTER(C.c_int64), _vp, C.POINTER(C.c_int64), _i32, _vp]
    L.fastegnn_check_finite.argtypes = [_vp, C.c_int64, _vp, C.c_int64, _vp, _vp]
    L.fastegnn_host_words_alloc.argtypes = [_i32, C.POINTER(C.POINTER(_i32))]
    L.fastegnn_host_words_free.argtypes = [C.POINTER(_i32)]
    L.fastegnn_zero_if_flagged.argtypes = [_vp, C.c_int64, _vp, _vp]
    L.fastegnn_spin_timeouts.argtypes = [_i32]
    L.fastegnn_gather_rows.argtypes = [_vp, _vp, C.c_int64, _i32, _vp, _vp]
    L.fastegnn_scatter_add_rows.argtypes = [_vp, _vp, C.c_int64, _i32, _vp, _vp]
    L.fastegnn_wg_slab_floats.restype = C.c_size_t
    L.fastegnn_pack_weights_all.argtypes = [C.POINTER(C.POINTER(LayerT)), _i32, _vp]
    L.fastegnn_wgrad_batch_open.argtypes = [C.POINTER(LayerT), _vp, C.POINTER(_vp)]
    L.fastegnn_wgrad_batch_close.argtypes = [_vp]
    _i64, _f = C.c_int64, C.c_float   # the wide path (include/fastegnn_hip.h "the WIDE path")
    L.fastegnn_wide_linear.argtypes = [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _f, _vp]
    L.fastegnn_wide_linear_dx.argtypes = [_vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _f, _vp]
    L.fastegnn_wide_linear_dw.argtypes = [_vp, _vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _f, _vp]
    L.fastegnn_wide_head_dx.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _f, _vp]
    L.fastegnn_wide_head_dw.argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _f, _i32, _f, _vp]
    L.fastegnn_wide_head_forward.argtypes = [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _i32, _f, _vp]
    L.fastegnn_wide_act.argtypes = [_vp, _i64, _i32, _f, _vp, _vp]
    L.fastegnn_wide_act_backward.argtypes = [_vp, _vp, _i64, _i32, _f, _vp, _vp]
    L.fastegnn_wide_gather_add.argtypes = [_vp, _vp, _i64, _i32, _vp, _vp, _vp]
    L.fastegnn_wide_scatter_add.argtypes = [_vp, _vp, _i64, _i32, _vp, _vp]
    L.fastegnn_wide_gather2.argtypes = [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp]
    L.fastegnn_wide_scatter_add_perm.argtypes = [_vp, _vp, _vp, _i64, _i32, _vp, _vp]
    L.fastegnn_wide_act_scatter.argtypes = [_vp, _vp, _i64, _i32, _i32, _f, _vp, _vp, _vp]
    L.fastegnn_wide_act_scatter_backward.argtypes = [_vp, _vp, _i64, _i32, _i32, _f, _vp, _vp, _vp, _vp]
    L.fastegnn_wide_rowscale.argtypes = [_vp, _vp, _i64, _i32, _vp, _vp]
    L.fastegnn_wide_rowdot.argtypes = [_vp, _vp, _i64, _i32, _vp, _vp]
    L.fastegnn_sizeof_layer.restype = C.c_size_t
    L.fastegnn_sizeof_graph.restype = C.c_size_t
    if L.fastegnn_sizeof_layer() != C.sizeof(LayerT) or L.fastegnn_sizeof_graph() != C.sizeof(GraphT):
        raise RuntimeError("fastegnn_amd: ctypes mirror of fastegnn_layer_t/fastegnn_graph_t is out of date")
    L.fastegnn_profile_enable.argtypes = [_i32]
    L.fastegnn_profile_kernels.restype = _i32
    L.fastegnn_profile_name.restype = C.c_char_p
    L.fastegnn_profile_name.argtypes = [_i32]
    L.fastegnn_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    for name in STAGE_FUNCS + ["fastegnn_layer_forward", "fastegnn_layer_backward"]:
        f = getattr(L, name)
        f.argtypes = [C.POINTER(LayerT), _vp]
        f.restype = C.c_int
    _libs[path] = L
    return L


STAGE_FUNCS = [
    "fastegnn_pack_weights", "fastegnn_node_pre_forward", "fastegnn_graph_xsum", "fastegnn_graph_pre_forward",
    "fastegnn_edge_forward", "fastegnn_virt_forward", "fastegnn_graph_post_forward",
    "fastegnn_graph_post_backward", "fastegnn_virt_backward", "fastegnn_graph_pre_backward",
    "fastegnn_edge_backward", "fastegnn_edge_col_reduce", "fastegnn_node_pre_backward",
]

# every symbol include/fastegnn_hip.h declares (checked by tests/test_abi_cpu.py)
EXPORTED = STAGE_FUNCS + [
    "fastegnn_last_error", "fastegnn_version", "fastegnn_wpack_floats", "fastegnn_wg_slab_floats", "fastegnn_wg_edge_floats", "fastegnn_wg_virt_floats", "fastegnn_wg_virt_floats_for", "fastegnn_backward_scratch_floats_for", "fastegnn_wg_node_floats", "fastegnn_backward_scratch_floats", "fastegnn_sizeof_layer", "fastegnn_sizeof_graph", "fastegnn_csr_tmp_bytes", "fastegnn_chunk_rows", "fastegnn_chunk_edges",
    "fastegnn_build_csr", "fastegnn_pad_params", "fastegnn_generic_activations", "fastegnn_f16_operands", "fastegnn_check_finite", "fastegnn_host_words_alloc", "fastegnn_host_words_free", "fastegnn_zero_if_flagged", "fastegnn_spin_timeouts", "fastegnn_permute_rows", "fastegnn_build_batch", "fastegnn_embed_forward",
    "fastegnn_embed_backward", "fastegnn_virtual_init", "fastegnn_virtual_init_backward",
    "fastegnn_layer_forward", "fastegnn_layer_backward", "fastegnn_selftest_gemm", "fastegnn_selftest_rm", "fastegnn_selftest_jreduce", "fastegnn_selftest_lane_sums", "fastegnn_selftest_wgrad", "fastegnn_selftest_wgrad_plan", "fastegnn_selftest_wgrad_guard", "fastegnn_selftest_stream", "fastegnn_selftest_chain", "fastegnn_selftest_chain_bf3",
    "fastegnn_augment_edge_attr", "fastegnn_loss_mse_mmd", "fastegnn_adam_step",
    "fastegnn_radius_graph_ws_bytes", "fastegnn_radius_graph_count", "fastegnn_radius_graph_fill",
    "fastegnn_cutoff_tmp_bytes", "fastegnn_cutoff_edges", "fastegnn_nbody_cutoff_edges",
    "fastegnn_profile_enable", "fastegnn_profile_kernels", "fastegnn_profile_name", "fastegnn_profile_collect",
    "fastegnn_comm_unique_id_bytes", "fastegnn_comm_unique_id", "fastegnn_comm_init", "fastegnn_comm_destroy", "fastegnn_comm_rank",
    "fastegnn_comm_world", "fastegnn_comm_all_reduce", "fastegnn_comm_all_gather", "fastegnn_comm_reduce_scatter",
    "fastegnn_comm_all_to_all_v", "fastegnn_gather_rows", "fastegnn_scatter_add_rows",
    "fastegnn_wgrad_batch_open", "fastegnn_wgrad_batch_close", "fastegnn_pack_weights_all",
    "fastegnn_wide_linear", "fastegnn_wide_linear_dx", "fastegnn_wide_linear_dw", "fastegnn_wide_head_dx", "fastegnn_wide_head_dw", "fastegnn_wide_head_forward", "fastegnn_wide_act", "fastegnn_wide_act_backward",
    "fastegnn_wide_gather_add", "fastegnn_wide_gather2", "fastegnn_wide_scatter_add", "fastegnn_wide_scatter_add_perm", "fastegnn_wide_act_scatter",
    "fastegnn_wide_act_scatter_backward", "fastegnn_wide_rowscale", "fastegnn_wide_rowdot",
]


def profile_collect():
    """-> {kernel name: (total ms, launches)} since the last collect (HIP events on the launch stream)."""
    L = lib()
    n = L.fastegnn_profile_kernels()
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    check(L.fastegnn_profile_collect(ms, cnt), "fastegnn_profile_collect")
    return {L.fastegnn_profile_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i] > 0}


def check(rc: int, what: str):
    if rc != 0:
        # every loaded build keeps its own last-error string (the generic-activation library is a second shared object)
        msgs = [f"[{os.path.basename(path)}] {m.decode()}" for path, m in
                ((path, L.fastegnn_last_error()) for path, L in _libs.items()) if m]
        raise RuntimeError(f"fastegnn_amd: {what} failed (code {rc}): {' | '.join(msgs) if msgs else '?'}")


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
