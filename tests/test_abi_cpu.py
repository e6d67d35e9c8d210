"""CPU: the C-ABI library loads and exports every symbol include/fastegnn_hip.h declares; the
ctypes mirrors match the C structs; the host-side module mirrors the reference interface
(state_dict layout, seeded init, error behaviour).  No compute calls (no GPU here)."""
import os
import re

import pytest
import torch

import fastegnn_amd
from fastegnn_amd import _lib as K
from tests.helpers import Golden, golden_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "fastegnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fastegnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = K.lib()
    names = _declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(K.EXPORTED) == set(names)


def test_generic_activation_library_exports_the_same_interface():
    """libfastegnn_hip_act.so (-DFE_ACT_GENERIC): same symbols, says what it is; the default library is the SiLU build."""
    A, L = K.lib(act=True), K.lib()
    assert A.fastegnn_generic_activations() == 1 and L.fastegnn_generic_activations() == 0
    assert not [n for n in _declared_symbols() if not hasattr(A, n)]
    assert A.fastegnn_sizeof_layer() == L.fastegnn_sizeof_layer()


def test_silu_library_refuses_other_activation_kinds_before_any_device_call():
    """The layer check runs on the host: the default library returns FASTEGNN_E_INVALID for any FASTEGNN_ACT_* kind but SiLU,
    the generic-activation build for kinds it does not know and for kinds combined with the bf16 operand mode."""
    import ctypes as C
    table = (C.c_void_p * K.P_COUNT)()
    def layer(flags):
        L = K.LayerT()
        L.N, L.B, L.C, L.ea, L.na, L.flags = 32, 1, 2, 2, 0, flags
        L.params = C.cast(table, C.c_void_p)
        return L
    lib, act = K.lib(), K.lib(act=True)
    assert lib.fastegnn_pack_weights(layer(K.ACT_RELU << K.F_ACT_SHIFT), None) == -1
    assert b"SiLU only" in lib.fastegnn_last_error()
    assert act.fastegnn_pack_weights(layer(9 << K.F_ACT_SHIFT), None) == -1
    assert b"unknown activation" in act.fastegnn_last_error()
    assert act.fastegnn_pack_weights(layer((K.ACT_GELU << K.F_ACT_SHIFT) | K.F_BF16), None) == -1
    assert b"bf16" in act.fastegnn_last_error()


def test_open_weight_gradient_batch_guards_its_operands():
    """the contractions of a layer run when the batch closes, so no later stage may write what a queued job
    still reads.  The layer driver declares every stage's writes to the open batch; here the overlap test itself (host-only:
    a job over rows [0, M) of G at `base` and of T at base + 64 M, probes around both)."""
    L = K.lib()
    M = 1000
    assert L.fastegnn_selftest_wgrad_guard(M, 0, 1) == -1                      # first float of G
    assert b"still has to read" in L.fastegnn_last_error()
    assert L.fastegnn_selftest_wgrad_guard(M, 64 * M - 1, 1) == -1             # last float of G
    assert L.fastegnn_selftest_wgrad_guard(M, 64 * M, 4) == -1                 # first floats of T
    assert L.fastegnn_selftest_wgrad_guard(M, 2 * 64 * M - 1, 1) == -1         # last float of T
    assert L.fastegnn_selftest_wgrad_guard(M, 2 * 64 * M, 1024) == 0           # just behind T
    assert L.fastegnn_selftest_wgrad_guard(M, -4096, 4096) == 0                # just in front of G
    assert L.fastegnn_selftest_wgrad_guard(M, -1, 2) == -1                     # straddles the start


def test_struct_mirrors_and_sizes():
    L = K.lib()
    import ctypes as C
    assert L.fastegnn_sizeof_layer() == C.sizeof(K.LayerT)
    assert L.fastegnn_sizeof_graph() == C.sizeof(K.GraphT)
    assert L.fastegnn_version() >= 100
    # fp32 + split images of the 34 + 2C matrices, row-major split images (64 rows x 144 B x 3 parts) of V2, WXV0, WXX0, W3c[c]
    # 34 fixed + 2 C images (fp32 + split) and 12 fixed (7 bf16-part + 5 f16x2 forms) + C row-major images
    assert L.fastegnn_wpack_floats(16) == (34 + 32) * (4096 + 4096 + 2048) + (12 + 16) * (3 * 64 * 144 // 4)
    assert L.fastegnn_profile_kernels() >= 15


def test_param_slots_follow_header_order():
    src = open(os.path.join(ROOT, "include", "fastegnn_hip.h")).read()
    slots = re.findall(r"FASTEGNN_P_([A-Z0-9_]+)\b(?:\s*=\s*0)?,", src)
    slots = [s for s in slots if s != "COUNT"]
    assert len(slots) == K.P_COUNT == len(K.PARAM_SLOTS)


@pytest.mark.parametrize("name", ["ragged3_allflags", "ragged3_nodeattr", "equiv10", "c16_two_graphs"])
def test_state_dict_layout_matches_reference(name):
    g = Golden(name)
    c = g.cfg
    m = fastegnn_amd.FastEGNN(c.node_feat_nf, c.node_attr_nf, c.edge_attr_nf, c.hidden_nf, c.virtual_channels,
                              n_layers=c.n_layers, residual=c.residual, attention=c.attention,
                              normalize=c.normalize, tanh=c.tanh, gravity=c.gravity)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g.params.keys())          # same keys, same order
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(g.params[k].shape), k
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g.params.items()}, strict=True)
    assert m.__class__.__name__ == "FastEGNN"                 # utils/train.py:51 dispatches on this


def test_seeded_init_equals_reference_init():
    # oracle/gen_goldens.py built 'ragged3_default_init' right after torch.manual_seed(3) without
    # touching the weights: same construction order => identical initial parameters
    g = Golden("ragged3_default_init")
    c = g.cfg
    torch.manual_seed(3)
    m = fastegnn_amd.FastEGNN(c.node_feat_nf, c.node_attr_nf, c.edge_attr_nf, c.hidden_nf, c.virtual_channels,
                              n_layers=c.n_layers)
    for k, v in m.state_dict().items():
        assert torch.equal(v, torch.from_numpy(g.params[k])), k


def test_error_behaviour():
    with pytest.raises(AssertionError):
        fastegnn_amd.FastEGNN(2, 0, 2, 64, 0)                 # models/FastEGNN.py:255
    assert fastegnn_amd.FastEGNN(2, 0, 2, 128, 3)._wide          # 64 < hidden_nf <= 256: the unfused wide path (fastegnn_amd/wide.py)
    with pytest.raises(NotImplementedError):
        fastegnn_amd.FastEGNN(2, 0, 2, 257, 3)                # beyond the wide path's range
    with pytest.raises(NotImplementedError):
        fastegnn_amd.FastEGNN(2, 0, 2, 128, 3, mlp_dtype=torch.bfloat16)   # the wide path is fp32 only
    assert fastegnn_amd.FastRF(2, 0, 2, 128, 3)._wide and fastegnn_amd.EGNN(2, 2, 2, 16, flat=True)._wide   # the siblings too
    # beyond the fused kernels' argument ceilings the wide path takes over as well (the reference has no such limits)
    assert fastegnn_amd.FastEGNN(9, 0, 2, 64, 3)._wide and fastegnn_amd.FastEGNN(2, 0, 8, 64, 3)._wide and fastegnn_amd.FastEGNN(2, 0, 2, 64, 65)._wide
    assert not fastegnn_amd.FastEGNN(8, 0, 7, 64, 64)._wide
    with pytest.raises(NotImplementedError):
        fastegnn_amd.FastEGNN(2, 0, 2, 64, 3, act_fn=torch.nn.Hardswish())      # not one of the eight kinds of the C ABI
    with pytest.raises(NotImplementedError):
        fastegnn_amd.FastEGNN(2, 0, 2, 32, 3, act_fn=torch.nn.Sigmoid())        # zero padding needs act_fn(0) = 0
    assert fastegnn_amd.FastEGNN(2, 0, 2, 64, 3, act_fn=torch.nn.LeakyReLU(0.2))._act == (K.ACT_LEAKY_RELU, 0.2)
    g = Golden("equiv10")
    m = fastegnn_amd.FastEGNN(1, 0, 1, 64, 3)
    kw, _, _ = g.model_kwargs()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(**kw)                                               # CPU tensors: never a silent fallback
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fastegnn_amd.FastEGNN(1, 0, 1, 128, 3)(**kw)          # the wide path likewise


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may use oracle/: no module under fastegnn_amd/
    imports it (statically checked on the import statements)."""
    import ast
    for root, _, files in os.walk(os.path.join(ROOT, "fastegnn_amd")):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(root, f)).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or ""]
                assert not any(m == "oracle" or m.startswith("oracle.") or m.startswith("tests") for m in mods), (f, mods)


def test_weight_gradient_slab_planning_never_exhausts_its_share():
    """the layer-wide weight-gradient batch (8 jobs of N rows, 5 of B*C rows, the edge stage's 2 x 256
    slabs; 4096 slabs, at most 384 per job) overflowed for N >= 393 k with B*C >= 30 k.  The contraction jobs are now
    planned together when the batch closes: if the share cannot hold the splits they ask for, all of them give up the
    same fraction (fewer, longer row ranges) instead of failing.  Host-only planning, no GPU."""
    import ctypes as C
    from fastegnn_amd import _lib
    L = _lib.lib()
    for N, BC in ((100000, 16), (393216, 32000), (600000, 64000), (1000000, 32), (4000000, 4096 * 64)):
        M = [N] * 3 + [BC] * 5 + [N] * 5          # virt (3 node-level jobs), graph_post + graph_pre (5), node_pre (5)
        nb = [1] * len(M)
        out = (C.c_int32 * len(M))()
        rc = L.fastegnn_selftest_wgrad_plan((C.c_int64 * len(M))(*M), (C.c_int32 * len(M))(*nb), len(M), 2, 256, 384, 4096, out)
        assert rc == 0, (N, BC, L.fastegnn_last_error())
        assert all(1 <= v <= 384 for v in out)
        assert sum(out) + 512 <= 4096
    # a share that cannot hold even one slab per job fails loudly instead of overrunning
    out = (C.c_int32 * 3)()
    rc = L.fastegnn_selftest_wgrad_plan((C.c_int64 * 3)(10 ** 6, 10 ** 6, 10 ** 6), (C.c_int32 * 3)(1, 1, 1), 3, 0, 0, 384, 2, out)
    assert rc != 0


def test_layer_list_pointer_table_does_not_outlive_its_tensors():
    """fastegnn_amd.sharded._LayerList caches the host pointer table of a layer's tensors.  It must not form a reference cycle:
    the gradient views of a backward have to die with the backward, or autograd's AccumulateGrad clones them instead of adopting
    the slices of the one flat buffer that dist.allreduce_gradients reduces in place (the 2-rank GPU tests caught exactly that)."""
    import gc
    import weakref
    from fastegnn_amd.sharded import _LayerList
    gc.collect()
    gc.disable()
    try:
        flat = torch.zeros(64)
        lst = _LayerList([flat[0:8], None, flat[8:16]] + [None] * (K.P_COUNT - 3))
        ref = weakref.ref(lst[0])
        tab = lst.ptab()
        assert tab is lst.ptab() and tab.arr[0] == flat.data_ptr() and tab.arr[1] is None
        del lst, tab
        assert ref() is None          # freed by reference counting alone (the collector is off)
    finally:
        gc.enable()


# (the wait states of inline-assembly blocks that write MFMA operands are checked on the DISASSEMBLY of the built libraries since
#  round 6: tests/test_isa_hazards_cpu.py, tools/isa_hazards.py)
