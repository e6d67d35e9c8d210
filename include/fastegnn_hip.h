/*
 * fastegnn_hip.h -- C ABI of libfastegnn_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for ONE path of GLAD-RUC/FastEGNN: the forward and backward of the
 * FastEGNN model (reference: models/FastEGNN.py:192-276 + autograd, utils/train.py:169).
 * The reference has no FFI for this path (it is a Python nn.Module calling ATen); the entry
 * points below are what a binding for it needs: plain pointers + sizes + a hipStream_t, the
 * caller allocates every buffer, no torch types, no global mutable state, re-entrant per
 * stream.  Each function returns 0 on success or a negative FASTEGNN_E_* code;
 * fastegnn_last_error() gives the message of the calling thread's last failure.
 *
 * All device arrays are fp32 row-major unless noted; indices are int32.
 * H (hidden_nf) is fixed to 64 in this build.
 *
 * One E_GCL_vel layer (models/FastEGNN.py:192-223) is evaluated as five stages; on one GPU
 * fastegnn_layer_forward/backward chain them, a sharded caller runs the stages itself and
 * exchanges the marked buffers between them (DESIGN.md "Multi-GPU").
 *
 *   forward                              backward (reverse order)
 *   S1 node_pre   h -> P, QX, A, heads    B1 node_pre_bwd
 *   S2 graph_pre  x,Z,Hv -> Bc            B3 graph_pre_bwd
 *   S3 edge       P,QX,CSR -> aggm,aggx   B2 edge_bwd (+ col-keyed reduce)
 *   S4 virt       ... -> h',x',pools      B4 virt_bwd
 *   S5 graph_post pools -> Z',Hv'         B5 graph_post_bwd
 */
#ifndef FASTEGNN_HIP_H
#define FASTEGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTEGNN_H 64
#define FASTEGNN_QX_LD 68 /* source-table row: Q[64] | x[3] | pad */

enum {
  FASTEGNN_OK = 0,
  FASTEGNN_E_INVALID = -1, /* bad argument (null pointer, unsupported size) */
  FASTEGNN_E_LAUNCH = -2,  /* HIP launch / runtime failure */
  FASTEGNN_E_NODEVICE = -3 /* no gfx950 device visible */
};

/* constructor flags of the reference model (models/FastEGNN.py:227-228, :11-12) */
enum {
  FASTEGNN_F_ATTENTION = 1,
  FASTEGNN_F_NORMALIZE = 2,
  FASTEGNN_F_TANH = 4,
  FASTEGNN_F_RESIDUAL = 8,
  FASTEGNN_F_GRAVITY = 16,
  FASTEGNN_F_COORDS_SUM = 32, /* coords_agg='sum' (default 'mean') */
  /* EGNN baseline layer (models/basic.py:285-320) on the same kernels: C = 0 (no virtual nodes),
   * edge_mlp.0 columns ordered [radial | h_row | h_col | edge_attr] (:313), coordinate head with bias,
   * aggregated coordinate message clamped to +-100 (:310), no residual on h, velocity head optional */
  FASTEGNN_F_EGNN = 64,
  /* FastRF reduced layer (models/FastRF.py:155-186) on the same kernels: node_mlp / node_mlp_virtual are
   * absent (h and the virtual features pass through unchanged, their parameter slots are null) and the
   * velocity scale is coord_mlp_vel(||vel||) with coord_mlp_vel.0.weight of shape [H,1] (:76-80,:139) */
  FASTEGNN_F_RF = 128,
  /* bf16 operand mode (BASELINE configs[2]; the reference has no reduced-precision mode): BOTH operands of every
   * 64-wide contraction -- the [64,64] weights and the 64-column blocks of edge_mlp.0 / edge_mlp_virtual.0 / node_mlp.0 /
   * node_mlp_virtual.0, in the forward, the input-gradient and the weight-gradient products -- are rounded to bf16
   * (round to nearest even) and multiplied on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Coordinates, radial /
   * vr / Gram terms and their weight columns, edge_attr / node_attr columns, the [1,64] heads, attention gates, biases,
   * activations, segment sums and pools stay fp32.  Mirror: oracle/factored.py with Config.bf16. */
  FASTEGNN_F_BF16 = 256,
  /* EGNN(norm=True) (models/basic.py:271-272): the 1x1 Gram feature of the message MLP is F.normalize'd -- r^2 / max(r^2, 1e-12),
   * i.e. 1 for every edge of non-zero length (gradient 0) and r^2 * 1e12 below (gradient 1e12).  With FASTEGNN_F_EGNN only. */
  FASTEGNN_F_EGNN_NORM = 512,
  /* Backward only.  By default the col-side adjoint of the edge stage -- d loss / d (Q | x)[col], the transpose of the
   * forward gather -- is scattered into g_QX_src with fp32 atomics (one coalesced 272-byte row per edge): g_QXe, the CSC
   * index of the graph (cscptr / csc_eid may then be NULL, also in fastegnn_build_csr) and fastegnn_edge_col_reduce's
   * kernel are not used, and the sums depend on the order the atomics land in (rounding-level run-to-run differences,
   * like torch's scatter_add_ on a GPU).  With this flag the rows are stored per edge in g_QXe [E,68] and summed in CSC
   * order by fastegnn_edge_col_reduce: that sum is then order-independent (the rest of the backward keeps its
   * rounding-level run-to-run differences: pools and ticket-ordered in-workgroup sums); 0.26 ms per step slower at cfg4 in
   * fp32, faster than the atomics with bf16 operands; 272 bytes of scratch per edge. */
  FASTEGNN_F_DETERMINISTIC = 1024,
  /* Backward, atomic mode only: fastegnn_edge_backward does NOT zero g_QX_src before it scatters into it -- an earlier
   * launch of the same layer over OTHER rows has already done so.  The sharded caller runs a rank's boundary rows (rows
   * with an edge from a ghost column) and its interior rows as two launches, so that the ghost rows' gradients travel
   * while the interior rows are still being computed (bits 11..14 hold the activation kind, below). */
  FASTEGNN_F_GQX_ACCUM = 32768,
  /* Forward: wpack already holds this layer's weight images (fastegnn_pack_weights_all packed every layer of the model in
   * one launch): fastegnn_layer_forward / fastegnn_pack_weights skip the per-layer pack launch. */
  FASTEGNN_F_WPACK_READY = 65536
};
/* Activation of every MLP (the reference's act_fn, models/FastEGNN.py:227): bits 11..14 of the flags hold one of the
 * FASTEGNN_ACT_* kinds, fastegnn_layer_t.act_param its parameter.  libfastegnn_hip.so is compiled for SiLU and rejects
 * any other kind (FASTEGNN_E_ARG); libfastegnn_hip_act.so (same sources, -DFE_ACT_GENERIC) evaluates all of them. */
#define FASTEGNN_F_ACT_SHIFT 11
#define FASTEGNN_F_ACT_MASK 15
enum {
  FASTEGNN_ACT_NONE = -1,        /* fastegnn_wide_linear*: no fused activation */
  FASTEGNN_ACT_SILU = 0,
  FASTEGNN_ACT_RELU = 1,
  FASTEGNN_ACT_LEAKY_RELU = 2,   /* act_param = negative_slope */
  FASTEGNN_ACT_TANH = 3,
  FASTEGNN_ACT_SIGMOID = 4,
  FASTEGNN_ACT_ELU = 5,          /* act_param = alpha */
  FASTEGNN_ACT_GELU = 6,         /* exact (erf) form */
  FASTEGNN_ACT_SOFTPLUS = 7      /* act_param = beta; threshold 20 as torch.nn.Softplus */
};

/* Per-layer parameter slots: the reference state_dict tensors of gcl_<i>, untouched
 * ([out,in] row-major as torch stores them).  models/FastEGNN.py:28-99. */
enum {
  FASTEGNN_P_EDGE0_W = 0, /* edge_mlp.0.weight          [64, 128+1+ea] */
  FASTEGNN_P_EDGE0_B,     /* edge_mlp.0.bias            [64] */
  FASTEGNN_P_EDGE2_W,     /* edge_mlp.2.weight          [64,64] */
  FASTEGNN_P_EDGE2_B,
  FASTEGNN_P_VIRT0_W,     /* edge_mlp_virtual.0.weight  [64, 128+1+C] */
  FASTEGNN_P_VIRT0_B,
  FASTEGNN_P_VIRT2_W,     /* edge_mlp_virtual.2.weight  [64,64] */
  FASTEGNN_P_VIRT2_B,
  FASTEGNN_P_ATT_W,       /* att_mlp.0.weight           [1,64]   (null unless attention) */
  FASTEGNN_P_ATT_B,       /* att_mlp.0.bias             [1] */
  FASTEGNN_P_ATTV_W,      /* att_mlp_virtual.0.weight   [1,64] */
  FASTEGNN_P_ATTV_B,
  FASTEGNN_P_CR0_W,       /* coord_mlp_r.0.weight       [64,64] */
  FASTEGNN_P_CR0_B,
  FASTEGNN_P_CR2_W,       /* coord_mlp_r.2.weight       [1,64] (no bias) */
  FASTEGNN_P_CRV0_W,      /* coord_mlp_r_virtual.*      */
  FASTEGNN_P_CRV0_B,
  FASTEGNN_P_CRV2_W,
  FASTEGNN_P_CVV0_W,      /* coord_mlp_v_virtual.*      */
  FASTEGNN_P_CVV0_B,
  FASTEGNN_P_CVV2_W,
  FASTEGNN_P_VEL0_W,      /* coord_mlp_vel.0.weight     [64,64] */
  FASTEGNN_P_VEL0_B,
  FASTEGNN_P_VEL2_W,      /* coord_mlp_vel.2.weight     [1,64] */
  FASTEGNN_P_VEL2_B,      /* [1] */
  FASTEGNN_P_GRAV0_W,     /* gravity_mlp.*  (null unless gravity) */
  FASTEGNN_P_GRAV0_B,
  FASTEGNN_P_GRAV2_W,
  FASTEGNN_P_GRAV2_B,
  FASTEGNN_P_NODE0_W,     /* node_mlp.0.weight          [64, 128+64*C+na] */
  FASTEGNN_P_NODE0_B,
  FASTEGNN_P_NODE2_W,     /* node_mlp.2.weight          [64,64] */
  FASTEGNN_P_NODE2_B,
  FASTEGNN_P_NODEV0_W,    /* node_mlp_virtual.0.weight  [64,128] */
  FASTEGNN_P_NODEV0_B,
  FASTEGNN_P_NODEV2_W,    /* node_mlp_virtual.2.weight  [64,64] */
  FASTEGNN_P_NODEV2_B,
  FASTEGNN_P_CR2_B,       /* coord head bias [1] (EGNN baseline only: coord_net.mlp.2.bias) */
  FASTEGNN_P_COUNT
};

/* Sorted graph produced by fastegnn_build_csr (replaces the implicit scatter indices of
 * unsorted_segment_sum/mean, models/FastEGNN.py:279-294). */
typedef struct {
  int32_t n_rows;          /* aggregation targets owned by this rank (== N on one GPU) */
  int32_t n_src;           /* rows of the source table the col indices address */
  int32_t n_edges;
  int32_t n_chunks;        /* edge-balanced row chunks (work items of the edge kernels) */
  const int32_t *rowptr;   /* [n_rows+1] */
  const int32_t *erow;     /* [E] row of sorted edge */
  const int32_t *col;      /* [E] source-table index of sorted edge */
  const int32_t *perm;     /* [E] sorted edge k == input edge perm[k] */
  const int32_t *cscptr;   /* [n_src+1] */
  const int32_t *csc_eid;  /* [E] sorted-edge ids grouped by col */
  const int32_t *chunk_row;/* [n_chunks+1] first row of each chunk */
} fastegnn_graph_t;

/* Everything one layer call touches.  "in"/"out" are w.r.t. the forward; the backward reads
 * the saved buffers and the g_* inputs and writes the g_* outputs (accumulating into grads). */
typedef struct {
  /* sizes / flags */
  int32_t N, B, C, ea, na, flags;
  float gravity[3];
  float epsilon;           /* models/FastEGNN.py:21 */
  float act_param;         /* parameter of the activation kind in the flags (FASTEGNN_ACT_*); 0 for SiLU */
  fastegnn_graph_t graph;
  const int32_t *batch;    /* [N] graph id per node, ascending */
  const int32_t *gptr;     /* [B+1] node range of each graph */
  const float *ea_sorted;  /* [E,ea] edge_attr in sorted-edge order */
  const float *vel;        /* [N,3] */
  const float *node_attr;  /* [N,na] or null */

  const float *const *params; /* HOST array [FASTEGNN_P_COUNT] of device pointers */
  float *const *grads;        /* HOST array [FASTEGNN_P_COUNT] of device pointers (+=), backward only */
  float *wpack;               /* packed MFMA weight images, fastegnn_wpack_floats(C) floats */

  /* layer inputs (saved for backward by the caller) */
  const float *h;          /* [N,64] */
  const float *x;          /* [N,3] */
  const float *Z;          /* [B,3,C]   virtual coordinates (reference layout) */
  const float *HvT;        /* [B,C,64]  virtual features, channel-major */
  /* layer outputs */
  float *h_out, *x_out, *Z_out, *HvT_out;

  /* stage products (saved for backward) */
  float *P;                /* [N,64]   S1 */
  float *QX;               /* [N,68]   S1: Q | x | pad.  (sharded: all-gathered into QX_src) */
  const float *QX_src;     /* [n_src,68] table the edge kernels gather from (== QX on one GPU) */
  float *A;                /* [N,64]   S1 */
  float *svel, *sgrav;     /* [N]      S1 */
  float *xsum;             /* [B,4]    S2: per-graph sum of x | node count (sharded: all-reduced) */
  float *Bc;               /* [B,C,64] S2 */
  float *aggm;             /* [N,64]   S3 */
  float *aggx;             /* [N,3]    S3 */
  float *npre;             /* [N,64]   S4: node_mlp pre-activation */
  float *poolV;            /* [B,C,64] S4 (sum over nodes; sharded: all-reduced) */
  float *poolX;            /* [B,3,C]  S4 (sum over nodes; sharded: all-reduced) */

  /* backward inputs: d loss / d layer outputs */
  const float *g_h_out, *g_x_out, *g_Z_out, *g_HvT_out;
  /* backward outputs: d loss / d layer inputs */
  float *g_h, *g_x, *g_Z, *g_HvT;
  float *g_vel;            /* [N,3] accumulated (+=), may be null */
  float *g_ea_sorted;      /* [E,ea] d loss / d edge_attr in sorted-edge order, accumulated (+=) over the layers; null: not wanted */
  float *g_node_attr;      /* [N,na] d loss / d node_attr, accumulated (+=) over the layers; null: not wanted */

  /* backward scratch (caller allocates; the shapes below, the wg_* sizes from fastegnn_wg_*_floats, the total from
   * fastegnn_backward_scratch_floats) */
  float *g_poolV, *g_poolX;   /* [B,C,64], [B,3,C] */
  float *g_Bc;                /* [B,C,64] (sharded: all-reduced before B3) */
  float *g_Zp;                /* [B,3,C]  partial from B4 (sharded: all-reduced) */
  float *g_xbar;              /* [B,3]    per-node share of d/d centroid */
  float *g_A, *g_P, *g_aggm;  /* [N,64] */
  float *g_aggx;              /* [N,3] */
  float *g_svel, *g_sgrav;    /* [N] */
  float *g_QXe;               /* [E,68]  per-edge d/d(Q|x) of the col side */
  float *g_QX_src;            /* [n_src,68] col-keyed sums (sharded: reduce-scattered) */
  float *g_QX;                /* [N,68]  this rank's slice of g_QX_src (== g_QX_src on one GPU) */
  float *g_xrow;              /* [N,3]   row-side d/dx of the edge stage */
  float *wg_edge;             /* [fastegnn_wg_edge_floats(E)] weight-gradient operands of the edge stage (none when the
                               * edge backward contracts them inside the workgroup: 4 floats then) */
  float *wg_virt;             /* [fastegnn_wg_virt_floats_for(N,C,flags)] weight-gradient operands of the virtual stage */
  float *wg_node;             /* [fastegnn_wg_node_floats(N,B,C)] node-level weight-gradient operands */
  float *wg_slab;             /* [fastegnn_wg_slab_floats()] partial 64x64 slabs of the weight-gradient GEMMs */
  /* Backward STAGE entry points only (fastegnn_layer_backward keeps its own): an open weight-gradient batch
   * (fastegnn_wgrad_batch_open) into which the stage queues its contraction jobs instead of contracting and reducing them
   * itself -- ONE contraction launch and ONE reduction launch per layer for a caller that drives the stages one by one
   * (the sharded path, which has collectives between them).  The operands a stage queues (its g_* outputs and its region
   * of wg_node) must stay untouched until fastegnn_wgrad_batch_close.  NULL: the stage finishes its own jobs. */
  void *wgrad_batch;
} fastegnn_layer_t;

/* ---- library ---- */
const char *fastegnn_last_error(void);
/* ABI revision: FASTEGNN_ABI_VERSION of the header the library was built from.  It changes whenever the layout of
 * fastegnn_layer_t / fastegnn_graph_t or the meaning of an argument changes (round 3 inserted act_param: 100 -> 101; round 4 appended wgrad_batch and added fastegnn_pack_weights_all / FASTEGNN_F_WPACK_READY: 103; the fastegnn_wide_* entry points: 104;
 * round 5: fastegnn_f16_operands / fastegnn_check_finite: 105; the fused activation arguments of fastegnn_wide_linear / _dx / _dw: 106;
 * round 6: fastegnn_host_words_alloc / _free, fastegnn_zero_if_flagged, fastegnn_check_finite writes 1 instead of OR-ing: 107).
 * A binding MUST compare it with the FASTEGNN_ABI_VERSION it was written against AND check fastegnn_sizeof_layer() /
 * fastegnn_sizeof_graph() against its own mirror of the descriptors before the first call (fastegnn_amd/_lib.py does). */
#define FASTEGNN_ABI_VERSION 107
int fastegnn_version(void);
/* floats of the packed weight-image buffer for C virtual channels */
size_t fastegnn_wpack_floats(int32_t C);
/* floats of the weight-gradient slab workspace (independent of the problem size) */
size_t fastegnn_wg_slab_floats(void);
size_t fastegnn_wg_edge_floats(int32_t E);
size_t fastegnn_wg_virt_floats(int32_t N, int32_t C);
size_t fastegnn_wg_node_floats(int32_t N, int32_t B, int32_t C);
/* wg_virt: three of the virtual stage's weight gradients are contracted inside the workgroup, so the workspace is ONE
 * [C][N+16][64] array plus constant-size part tiles and consumer scratch (~36 MB at C = 16; 4 floats at C = 0).  Until
 * round 4 the FastRF / EGNN wirings kept five arrays and the size depended on the flags; every wiring has the one form
 * now and the _for variant (kept for ABI stability) ignores `flags`. */
size_t fastegnn_wg_virt_floats_for(int32_t N, int32_t C, int32_t flags);
/* floats of all backward scratch arrays of fastegnn_layer_t together (g_poolV ... wg_slab), each array rounded up to
 * a multiple of 4 floats so that one allocation can be carved into 16-byte aligned pieces */
size_t fastegnn_backward_scratch_floats(int32_t N, int32_t E, int32_t n_src, int32_t B, int32_t C);
size_t fastegnn_backward_scratch_floats_for(int32_t N, int32_t E, int32_t n_src, int32_t B, int32_t C, int32_t flags);
/* A weight-gradient batch shared by the backward stage calls of ONE layer (fastegnn_layer_t.wgrad_batch).  open: L gives
 * wg_slab (the batch takes the lower half of the slab workspace, as fastegnn_layer_backward's does) and the operand mode;
 * the stages must run on `stream`.  close: contracts and reduces everything queued (two launches), then frees the handle --
 * also call it after a failed stage.  Host-side objects only: capturable. */
int fastegnn_wgrad_batch_open(const fastegnn_layer_t *L, void *stream, void **batch);
int fastegnn_wgrad_batch_close(void *batch);
/* struct sizes, so that a foreign-language binding can verify its mirror of the descriptors */
size_t fastegnn_sizeof_layer(void);
size_t fastegnn_sizeof_graph(void);

/* ---- graph preprocessing (COO int64, any order -> row-sorted CSR + col-keyed index) ----
 * edge_index: device int64 [2,E] as the reference passes it (models/FastEGNN.py:204).
 * Rows must lie in [row_begin, row_begin+n_rows) and are stored relative to row_begin; cols
 * address the source table [0,n_src).  All outputs caller-allocated:
 * rowptr[n_rows+1], erow/col/perm/csc_eid[E], cscptr[n_src+1], chunk_row[fastegnn_chunk_rows(E)]
 * (row boundaries nearest to every 32nd edge: the edge kernels give each wave a contiguous run of them);
 * tmp: fastegnn_csr_tmp_bytes(E, n_rows, n_src) bytes.  *n_chunks is a host int.
 * The index is the STABLE sort's (edges of a row keep the caller's order): built with rocPRIM radix sorts, or -- edge lists of
 * at most 600 000 edges with at most 64 edges per id on average -- by counting (count, scan, place, rank inside the id), which gives
 * the same arrays element for element; FASTEGNN_CSR_SORT=radix|count in the environment forces one form.
 * The edge stages address their tables with 32-bit offsets: n_rows * 68 and n_src * 68 must stay below 2^30
 * (15.7 M nodes) and E * 8 below 2^30 (134 M edges); fastegnn_edge_forward / _backward return
 * FASTEGNN_E_INVALID beyond that. */
size_t fastegnn_csr_tmp_bytes(int32_t E, int32_t n_rows, int32_t n_src);
size_t fastegnn_chunk_rows(int32_t E);
int32_t fastegnn_chunk_edges(void);   /* edges per row chunk: chunk k owns the rows whose first edge lies in [k*T, (k+1)*T) */
int fastegnn_build_csr(const int64_t *edge_index, int32_t E, int32_t row_begin, int32_t n_rows,
                       int32_t n_src, int32_t *rowptr, int32_t *erow, int32_t *col, int32_t *perm,
                       int32_t *cscptr, int32_t *csc_eid, int32_t *chunk_row, int32_t *n_chunks,
                       void *tmp, size_t tmp_bytes, void *stream);
/* out[k,:] = in[perm[k],:]   (edge_attr into sorted order) */
int fastegnn_permute_rows(const float *in, const int32_t *perm, int32_t E, int32_t width, float *out,
                          void *stream);
/* hidden_nf < 64: the kernels work on 64-wide tiles, a narrower model runs zero-padded (a zero row of a Linear gives a
 * zero pre-activation, SiLU(0) = 0, a zero column ignores its input: the padded model computes the reference's function,
 * models/FastEGNN.py:28-99 with hidden_nf = h).  One descriptor per parameter: `src` [rows, cols] row-major is copied into
 * `dst` [rows_dst, cols_dst] (zero elsewhere); the first `nblk` column blocks of blk[i] source columns each -- the
 * hidden-sized pieces of the reference's torch.cat inputs, FastEGNN.py:104,114,157,171 -- are widened to
 * blk[i] / h * 64 columns (zeros appended to the block), the remaining columns follow unchanged; `lead` source columns
 * in front of the first block are copied as they are (the radial column of the EGNN baseline's message MLP, basic.py:313).
 * reverse != 0 runs the
 * adjoint: src[r, c] = dst[r, map(c)] (the gradient of the narrow parameter is the matching slice of the padded one).
 * `desc` is a HOST array; at most 64 descriptors travel per launch (kernel arguments, capturable into a HIP graph). */
typedef struct {
  const float *src;        /* reverse: written */
  float *dst;              /* reverse: read */
  int32_t rows, cols;      /* of src */
  int32_t rows_dst, cols_dst;
  int32_t nblk;
  int32_t blk[3];
  int32_t lead;            /* unchanged columns in front of the blocks */
  int32_t reserved;
} fastegnn_pad_desc_t;
int fastegnn_pad_params(const fastegnn_pad_desc_t *desc, int32_t n, int32_t h, int32_t reverse, void *stream);
/* 1 when the library evaluates every FASTEGNN_ACT_* kind (libfastegnn_hip_act.so), 0 for the SiLU-only build */
int fastegnn_generic_activations(void);
/* 1 if the fp32-grade products of this build run on 2-part fp16 splits (operands must stay below 65 504 in magnitude: the default
 * library), 0 if on 3-part bf16 splits with fp32's exponent range (libfastegnn_hip_x3.so / _act_x3.so).  ABI revision 105. */
int fastegnn_f16_operands(void);
/* Range guard of the f16x2 build (the reference is plain fp32, models/FastEGNN.py:102-119: it has no such limit): *flag = 1 if any
 * of a[0..na) / b[0..nb) is not finite (a plain store of the constant: `flag` may be device memory OR a host-mapped word of
 * fastegnn_host_words_alloc, which the host then reads without any synchronisation).  One capturable launch on `stream`.
 * Either array may be NULL with its count 0.  ABI revision 105; store instead of atomic OR since 107. */
int fastegnn_check_finite(const float *a, int64_t na, const float *b, int64_t nb, int32_t *flag, void *stream);
/* n zero-initialised int32 words of pinned, host-MAPPED, coherent memory: kernels write them through the same address (the guard
 * launch above), the host polls them with plain loads -- no stream synchronisation, no copy, valid inside a captured HIP graph as
 * well.  The reference's only host synchronisation on this path is data_batch[-1].item() (models/FastEGNN.py:267), which the module
 * does not need; this keeps the range guard from adding one.  ABI revision 107. */
int fastegnn_host_words_alloc(int32_t n, int32_t **words);
int fastegnn_host_words_free(int32_t *words);
/* buf[0..n) = 0 if *flag != 0 (flag: device or host-mapped word).  The backward of a forward whose outputs left the f16x2 range
 * hands ZERO parameter gradients to the optimizer instead of NaNs (the host may not have seen the flag yet).  ABI revision 107. */
int fastegnn_zero_if_flagged(float *buf, int64_t n, const int32_t *flag, void *stream);
/* The channel-phased virtual backward hands work between the waves of a workgroup through LDS flags; since round 6 every wait on such a
 * flag is bounded (~0.3 s).  Returns how many waits have given up since the last reset -- 0 unless the hand-off protocol has a bug (the
 * launch then finished with wrong results instead of hanging the device); < 0: the query failed.  SYNCHRONISES the device: for tests and
 * post-mortems, not for the step.  ABI revision 107. */
int fastegnn_spin_timeouts(int reset);
/* batch int64 [N] (ascending) -> batch int32 [N], gptr int32 [B+1] */
int fastegnn_build_batch(const int64_t *batch64, int32_t N, int32_t B, int32_t *batch, int32_t *gptr,
                         void *stream);

/* ---- model prologue / epilogue (models/FastEGNN.py:267-271) ---- */
/* h = node_feat @ W^T + b   (embedding_in) */
int fastegnn_embed_forward(const float *node_feat, int32_t N, int32_t nf, const float *W, const float *b,
                           float *h, void *stream);
/* gW += g_h^T node_feat, gb += colsum g_h, g_node_feat = g_h W (nullable) */
int fastegnn_embed_backward(const float *node_feat, const float *g_h, int32_t N, int32_t nf, const float *W,
                            float *gW, float *gb, float *g_node_feat, void *stream);
/* HvT[b,c,:] = virtual_node_feat[0,:,c]  (the .repeat(B,1,1) of :268, channel-major) */
int fastegnn_virtual_init(const float *vnf, int32_t B, int32_t C, float *HvT, void *stream);
/* g_vnf[0,h,c] += sum_b g_HvT[b,c,h] */
int fastegnn_virtual_init_backward(const float *g_HvT, int32_t B, int32_t C, float *g_vnf, void *stream);

/* ---- one layer, staged ---- */
int fastegnn_pack_weights(const fastegnn_layer_t *L, void *stream);
/* The weight images of n layers of ONE model (same C, ea, na, flags; each descriptor needs params, wpack and the sizes) in one
 * launch: a layer's images are 34 + 2C small workgroups, and four launches of them cost four launch latencies in front of a step
 * that is otherwise 74 launches.  Set FASTEGNN_F_WPACK_READY in the flags of the layer calls that follow. */
int fastegnn_pack_weights_all(const fastegnn_layer_t *const *layers, int32_t n, void *stream);
int fastegnn_node_pre_forward(const fastegnn_layer_t *L, void *stream);   /* S1 */
int fastegnn_graph_xsum(const fastegnn_layer_t *L, void *stream);         /* S2a: local xsum */
int fastegnn_graph_pre_forward(const fastegnn_layer_t *L, void *stream);  /* S2b: xsum -> Bc */
int fastegnn_edge_forward(const fastegnn_layer_t *L, void *stream);       /* S3 */
int fastegnn_virt_forward(const fastegnn_layer_t *L, void *stream);       /* S4 (zeroes + fills pools) */
int fastegnn_graph_post_forward(const fastegnn_layer_t *L, void *stream); /* S5 */

int fastegnn_graph_post_backward(const fastegnn_layer_t *L, void *stream); /* B5 */
int fastegnn_virt_backward(const fastegnn_layer_t *L, void *stream);       /* B4 */
int fastegnn_graph_pre_backward(const fastegnn_layer_t *L, void *stream);  /* B3 */
int fastegnn_edge_backward(const fastegnn_layer_t *L, void *stream);       /* B2a: per-edge */
int fastegnn_edge_col_reduce(const fastegnn_layer_t *L, void *stream);     /* B2b: g_QXe -> g_QX_src */
int fastegnn_node_pre_backward(const fastegnn_layer_t *L, void *stream);   /* B1 */

/* ---- one layer, whole (single GPU): pack + S1..S5 / B5..B1 on the stream ---- */
int fastegnn_layer_forward(const fastegnn_layer_t *L, void *stream);
int fastegnn_layer_backward(const fastegnn_layer_t *L, void *stream);

/* ---- training-step closure (the harness code that brackets the hot path, utils/train.py) ----
 * out[e,:] = [edge_attr[e,:k] | ||loc[row_e]-loc[col_e]||]                      (utils/train.py:41-43) */
int fastegnn_augment_edge_attr(const int64_t *edge_index, const float *loc, const float *edge_attr, int32_t E,
                               int32_t k, float *out, void *stream);
/* loss2[0] = MSE(loc_pred,loc_t) + weight*(l_vv - l_rv), loss2[1] = MSE      (utils/train.py:104-165, kernel :17-20);
 * sample_nodes int32 [B,S]: absolute ids of the sampled real nodes of each graph; writes d loss / d loc_pred into
 * g_loc [N,3] and d loss / d virtual_node_loc into g_vloc [B,3,C]. */
int fastegnn_loss_mse_mmd(const float *loc_pred, const float *loc_t, const float *vloc, const int32_t *sample_nodes,
                          int32_t N, int32_t B, int32_t C, int32_t S, float sigma, float weight, float *loss2,
                          float *g_loc, float *g_vloc, void *stream);
/* torch.optim.Adam step (no amsgrad, L2 weight decay; main_nbody.py:137) over n_tensors tensors given as HOST arrays
 * of device pointers; a tensor whose grads[i] is null is skipped entirely, as torch.optim.Adam skips parameters whose
 * .grad is None (no weight decay, no moment update); step counts from 1. */
int fastegnn_adam_step(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                       const int64_t *numel, int32_t n_tensors, int32_t step, float lr, float beta1, float beta2,
                       float eps, float weight_decay, void *stream);

/* ---- graph construction on device (datasets/simulation/dataset.py:80,96-101) ----
 * radius graph without self loops: all ordered pairs (i, j != i) with |x_i - x_j|^2 <= r^2 (fp32, each
 * operation rounded separately), grouped by i with j ascending.  Two passes so that the caller allocates:
 * _count fills the workspace (fastegnn_radius_graph_ws_bytes(N) bytes) and returns the edge count in
 * *n_edges (host; synchronises the stream), _fill writes edge_index int64 [2,E] and dist [E]. */
size_t fastegnn_radius_graph_ws_bytes(int32_t N);
int fastegnn_radius_graph_count(const float *loc, int32_t N, float r, void *ws, size_t ws_bytes, int64_t *n_edges,
                                void *stream);
int fastegnn_radius_graph_fill(const float *loc, int32_t N, float r, void *ws, size_t ws_bytes, int64_t n_edges,
                               int64_t *edge_index, float *dist, void *stream);
/* keep the `keep` shortest edges (stable: ties keep their order), in ascending length like the reference's
 * cutoff_edge; tmp: fastegnn_cutoff_tmp_bytes(E) bytes */
size_t fastegnn_cutoff_tmp_bytes(int64_t E);
int fastegnn_cutoff_edges(const int64_t *edge_index, const float *dist, int64_t E, int64_t keep, int64_t *edge_index_out,
                          float *dist_out, void *tmp, size_t tmp_bytes, void *stream);

/* N-body systems (datasets/nbody/dataset.py:102-113): for each of S systems of n <= 128 particles (loc [S,n,3]) the k
 * shortest ordered pairs (i, j != i) of the complete graph in ascending length (fp32 distance, each operation rounded
 * separately; equal lengths in ascending i*n+j): edge_index int64 [S,2,k] (particle ids within the system), dist [S,k]. */
int fastegnn_nbody_cutoff_edges(const float *loc, int32_t S, int32_t n, int32_t k, int64_t *edge_index, float *dist,
                                void *stream);

/* ---- exchange steps of the sharded path (the reference has no distributed code) ----
 * RCCL collectives behind the C ABI: every call is enqueued on `stream` (ordered with the stage kernels, capturable
 * into a HIP graph) and returns at once.  RCCL is bound at run time (dlopen; FASTEGNN_E_NODEVICE when absent).
 * Bring-up: rank 0 calls fastegnn_comm_unique_id, the caller distributes the fastegnn_comm_unique_id_bytes() bytes by
 * any out-of-band means, every rank calls fastegnn_comm_init.  One communicator per process and device. */
typedef struct fastegnn_comm fastegnn_comm_t;
int32_t fastegnn_comm_unique_id_bytes(void);
int fastegnn_comm_unique_id(void *id);
int fastegnn_comm_init(fastegnn_comm_t **comm, const void *id, int32_t rank, int32_t world);
int fastegnn_comm_destroy(fastegnn_comm_t *comm);
int32_t fastegnn_comm_rank(const fastegnn_comm_t *comm);
int32_t fastegnn_comm_world(const fastegnn_comm_t *comm);
/* buf <- sum over ranks (xsum [B,4]; poolV|poolX; g_Bc|g_Zp; the flat parameter-gradient buffer) */
int fastegnn_comm_all_reduce(fastegnn_comm_t *comm, float *buf, size_t n, void *stream);
/* QX [Npad,68] of every rank -> QX_src [W*Npad,68] ("allgather" table exchange) and its transpose */
int fastegnn_comm_all_gather(fastegnn_comm_t *comm, const float *in, float *out, size_t n_per_rank, void *stream);
int fastegnn_comm_reduce_scatter(fastegnn_comm_t *comm, const float *in, float *out, size_t n_per_rank, void *stream);
/* halo exchange: send_rows[r] rows of row_floats floats go to rank r (from `send`, in rank order), recv_rows[r] rows
 * arrive from rank r (into `recv`, in rank order); the backward swaps the two count arrays.  HOST arrays [world]. */
int fastegnn_comm_all_to_all_v(fastegnn_comm_t *comm, const float *send, const int64_t *send_rows, float *recv,
                               const int64_t *recv_rows, int32_t row_floats, void *stream);
/* ghost-row pack / unpack of the halo exchange: out[r,:] = table[ids[r],:] (width % 4 == 0);  table[ids[r],:] += rows[r,:] */
int fastegnn_gather_rows(const float *table, const int64_t *ids, int64_t n, int32_t width, float *out, void *stream);
int fastegnn_scatter_add_rows(float *table, const int64_t *ids, int64_t n, int32_t width, const float *rows, void *stream);

/* ---- the WIDE path: 64 < hidden_nf <= 256 (models/FastEGNN.py:28-99 takes any hidden_nf; main_nbody.py:27 --dim_hidden) ----
 * The fused stage kernels above are built on 64-wide tiles.  A wider model runs unfused: the op sequence of
 * models/FastEGNN.py:102-223 with every hidden-sized tensor op as ONE of the launches below (fastegnn_amd/wide.py; autograd
 * composes the backward from the _dx / _dw / _backward entry points).  fp32-grade arithmetic (the GEMMs as bf16x3 splits on the
 * matrix pipe, csrc/wide_gemm.h), row-major contiguous operands, int64 indices as the reference's edge_index / data['batch']
 * hold them.  ABI revision 104; revision 106 added the fused activation arguments of the three linear entry points:
 * act_kind = FASTEGNN_ACT_* makes X the PRE-activation of the Linear's input (nn.Sequential(Linear, act, Linear),
 * models/FastEGNN.py:41-99: act(X) is formed in the kernel and never stored), FASTEGNN_ACT_NONE takes X as is.
 *   linear      out[M,O] = (base ? base : 0) + act(X)[M,K] . W[:, c0:c0+K]^T + bias -- nn.Linear; a Linear over a torch.cat of
 *               inputs is the sum of these calls over the weight's column blocks (W row stride ldw), chained through `base`
 *   linear_dx   dX[M,K] (+)= (G[M,O] . W[:, c0:c0+K]) * act'(Z[M,K])   -- Z NULL: no activation factor; with Z the result is the
 *               gradient of the pre-activation Z that `linear` took
 *   linear_dw   dW[:, c0:c0+K] += G^T act(X),  db += column sums of G (either may be NULL); fp32 atomics over row ranges
 *   head_dx / head_dw   the backward of the first Linear of a scalar head s = act(X W1^T + b1) . w2^T (coord_mlp_*, gravity_mlp:
 *               models/FastEGNN.py:55-99) from the head's output gradient gs[M]: G[m,o] = gs[m] w2[o] act'(Zc[m,o]) is formed in
 *               the kernels from the stored pre-activation Zc and never written; O (hidden width) a multiple of 4, O and K >= 9;
 *               head_dw also returns the second Linear's weight gradient dw2[o] += sum_m gs[m] act(Zc[m,o]) (may be NULL)
 *   head_forward  Zc = act_x(X) W1^T + b1 (stored) and s = act(Zc) . w2 + b2: with 9 .. 128 hidden units s comes out of the first
 *               GEMM's accumulators (no pass over Zc), otherwise from a second launch
 *   act         y = act_fn(z), kind = FASTEGNN_ACT_*, p = its parameter;  act_backward  dz = dy * act_fn'(z)
 *   gather_add  out[m,:] = (base ? base[m,:] : 0) + X[idx[m],:]      -- node_feat[row], virtual_node_feat[data_batch]
 *   gather2     out[m,:] = (base) + P[i1[m],:] + (Q ? Q[i2[m],:] : 0) + feat[m,0:nf] . Wf[:, c0:c0+nf]^T  (feat may be NULL, nf <= 8)
 *               -- the first Linear of edge_model over cat[h[row], h[col], radial, edge_attr] (models/FastEGNN.py:102-108) in one
 *               write-only pass, given the node-sized products P and Q; edge_mode_virtual (:111-119) in the same form
 *   scatter_add table[idx[m],:] += rows[m,:]                         -- unsorted_segment_sum / global_mean_pool sums (atomics;
 *               runs of equal targets are summed in registers first)
 *   act_scatter y = act_fn(z) stored AND table[idx[m],:] += y[m,:] in one pass (edge_mlp's output and its segment sum for node_model,
 *               models/FastEGNN.py:108, 156; edge_mlp_virtual's and node_model_virtual's pool, :119, 170);
 *               act_scatter_backward  dz = ((g_y ? g_y : 0) + g_table[idx]) * act_fn'(z)
 *   scatter_add_perm  table[idx_sorted[m],:] += rows[perm[m],:]      -- the same for an index in any order, given the permutation
 *               that sorts it (idx_sorted = idx[perm], once per graph): the edge COLUMN sums of the backward as runs
 *   rowscale    Y[m,:] = X[m,:] * s[m];  rowdot  out[m] = <A[m,:], B[m,:]>   -- gates, 1/count of the segment means */
int fastegnn_wide_linear(const float *X, int64_t M, int32_t K, const float *W, int32_t ldw, int32_t c0, const float *bias,
                         const float *base, float *out, int32_t O, int32_t act_kind, float act_p, void *stream);
int fastegnn_wide_linear_dx(const float *G, int64_t M, int32_t O, const float *W, int32_t ldw, int32_t c0, int32_t K, float *dX,
                            int32_t accumulate, const float *Z, int32_t act_kind, float act_p, void *stream);
int fastegnn_wide_linear_dw(const float *G, const float *X, int64_t M, int32_t O, int32_t K, float *dW, int32_t ldw, int32_t c0,
                            float *db, int32_t act_kind, float act_p, void *stream);
int fastegnn_wide_head_dx(const float *gs, const float *w2, const float *Zc, int64_t M, int32_t O, const float *W, int32_t ldw,
                          int32_t c0, int32_t K, float *dX, int32_t accumulate, int32_t kind, float p, void *stream);
int fastegnn_wide_head_dw(const float *gs, const float *w2, const float *Zc, const float *X, int64_t M, int32_t O, int32_t K, float *dW,
                          int32_t ldw, int32_t c0, float *db, float *dw2, int32_t kind, float p, int32_t x_kind, float x_p, void *stream);
int fastegnn_wide_head_forward(const float *X, int64_t M, int32_t K, const float *W1, int32_t ldw, int32_t c0, const float *b1,
                               const float *w2, const float *b2, float *Zc, float *s, int32_t O, int32_t kind, float p, int32_t x_kind,
                               float x_p, void *stream);
int fastegnn_wide_act(const float *z, int64_t n, int32_t kind, float p, float *y, void *stream);
int fastegnn_wide_act_backward(const float *z, const float *dy, int64_t n, int32_t kind, float p, float *dz, void *stream);
int fastegnn_wide_gather_add(const float *X, const int64_t *idx, int64_t M, int32_t W, const float *base, float *out, void *stream);
int fastegnn_wide_gather2(const float *P, const int64_t *i1, const float *Q, const int64_t *i2, const float *feat, int32_t nf,
                          const float *Wf, int32_t ldw, int32_t c0, const float *base, float *out, int64_t M, int32_t W, void *stream);
int fastegnn_wide_scatter_add(float *table, const int64_t *idx, int64_t M, int32_t W, const float *rows, void *stream);
int fastegnn_wide_act_scatter(const float *z, const int64_t *idx, int64_t M, int32_t W, int32_t kind, float p, float *y, float *table,
                              void *stream);
int fastegnn_wide_act_scatter_backward(const float *z, const int64_t *idx, int64_t M, int32_t W, int32_t kind, float p, const float *g_y,
                                       const float *g_table, float *dz, void *stream);
int fastegnn_wide_scatter_add_perm(float *table, const int64_t *idx_sorted, const int64_t *perm, int64_t M, int32_t W, const float *rows,
                                   void *stream);
int fastegnn_wide_rowscale(const float *X, const float *s, int64_t M, int32_t W, float *Y, void *stream);
int fastegnn_wide_rowdot(const float *A, const float *B, int64_t M, int32_t W, float *out, void *stream);

/* ---- per-kernel timing with HIP events recorded on the launch stream (bench.py) ----
 * enable(1) brackets every kernel launch of this library with two events; collect() waits for
 * them and returns, per kernel id in [0, fastegnn_profile_kernels()), the summed duration in ms
 * and the launch count since the previous collect().  Not thread-safe; off by default. */
int fastegnn_profile_enable(int32_t on);
int32_t fastegnn_profile_kernels(void);
const char *fastegnn_profile_name(int32_t id);
int fastegnn_profile_collect(double *total_ms, int64_t *launches);

/* ---- diagnostics (used by tests/ only) ----
 * Y[j][o] = sum_k A[o][k] X[j][k] for one 16-row tile through the MFMA image path (A = W or W^T,
 * W 64x64 row-major);  dW += G^T T, db += colsum(G) over M rows of 64. */
int fastegnn_selftest_gemm(const float *W, const float *X, float *Y, int32_t transposed, void *stream);
/* same through a row-major split image in LDS (one copy serves W and W^T: ds_read_b64 / ds_read_b64_tr_b16);
 * mode 0: bf16x3 products, 1: one bf16 product of the rounded operands */
int fastegnn_selftest_rm(const float *W, const float *X, float *Y, int32_t transposed, int32_t mode, void *stream);
/* transposing tile sum of the backward kernels (common.h jreduce16): X one 16x64 tile; out[16 q + j] = sum over the 16
 * rows of X[row][16 (j >> 2) + 4 q + (j & 3)] */
int fastegnn_selftest_jreduce(const float *X, float *out, void *stream);
/* X [64] -> out [128]: the kernels' cross-lane sums of one wave: out[l] = sum over the lanes l % 16 + 16 q (v_permlane16_swap /
 * v_permlane32_swap), out[64 + l] = sum over the 16 lanes of l's row (DPP rotations) */
int fastegnn_selftest_lane_sums(const float *X, float *out, void *stream);
/* `iters` dependent 64x64 MFMA layers per wave (mode bit0: SiLU between layers, bit1: image from
 * global memory instead of LDS); out receives one 16x64 tile.  Calibrates the MFMA building block. */
int fastegnn_selftest_chain(const float *wimg, float *out, int32_t iters, int32_t mode, int32_t waves, int32_t grid,
                            void *stream);
/* same chain on the 3-way bf16 split path (W 64x64 row-major fp32, X one 16x64 tile); mode bit2: one
 * layer, raw output (accuracy check against fp64). */
int fastegnn_selftest_chain_bf3(const float *W, const float *X, float *out, int32_t iters, int32_t mode, int32_t waves,
                                int32_t grid, void *stream);
int fastegnn_selftest_wgrad(const float *G, const float *T, int32_t M, float *dW, float *db, float *slab,
                            void *stream);
/* host-only: slab planning of a weight-gradient batch (n_jobs jobs of M[k] rows x nb[k] slices, then n_slab_jobs jobs of
 * slabs_each caller-written slabs) under a split limit and a slab share; nsplit_out[k] = partial slabs per slice of job k */
int fastegnn_selftest_wgrad_plan(const int64_t *M, const int32_t *nb, int32_t n_jobs, int32_t n_slab_jobs, int32_t slabs_each,
                                 int32_t max_split, int32_t slab_cap, int32_t *nsplit_out);
/* Host-only self-test of the open batch's overwrite guard (a stage must not write what a queued, not yet contracted job reads) */
int fastegnn_selftest_wgrad_guard(int64_t M, int64_t probe_off, int64_t probe_n);
/* HBM streaming calibration: mode 0 reads src (n_floats, multiple of 4), 1 copies src -> dst, 2 writes dst */
int fastegnn_selftest_stream(const float *src, float *dst, size_t n_floats, int32_t mode, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTEGNN_HIP_H */
