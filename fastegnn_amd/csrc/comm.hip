// fastegnn_comm_*: the exchange steps of the sharded path behind the C ABI, on RCCL.
// Every call is enqueued on the caller's stream (stream-ordered with the stage kernels, capturable into a HIP graph, no
// host synchronisation) and returns at once.  RCCL is bound at run time (dlopen; the copy torch has already loaded is
// reused), so libfastegnn_hip.so itself has no link-time dependency on it and loads on a box without RCCL.
// The reference has no distributed code: nothing to cite but the contract of section 8e.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "kernels.h"

namespace {

// the part of the NCCL / RCCL API this file uses (rccl.h: stable C ABI)
typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
enum { NCCL_FLOAT32 = 7, NCCL_SUM = 0 };
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(nccl_uid_t *) = nullptr;
  int (*CommInitRank)(nccl_comm_t *, int, nccl_uid_t, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *n : names)   // a copy that is already in the process (torch's) first
    if ((g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!g_rccl.h)
    for (const char *n : names)
      if ((g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!g_rccl.h) return;
#define SYM(field, name) *reinterpret_cast<void **>(&g_rccl.field) = dlsym(g_rccl.h, name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(AllGather, "ncclAllGather");
  SYM(ReduceScatter, "ncclReduceScatter");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.AllGather &&
              g_rccl.ReduceScatter && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
}

int rccl_ready() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok) {
    fe::set_error("fastegnn_comm: librccl.so could not be loaded (or lacks a symbol)");
    return FASTEGNN_E_NODEVICE;
  }
  return FASTEGNN_OK;
}
int rccl_check(int rc, const char *what) {
  if (rc == 0) return FASTEGNN_OK;
  fe::set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
  return FASTEGNN_E_LAUNCH;
}

}  // namespace

struct fastegnn_comm {
  nccl_comm_t comm;
  int rank, world;
};

extern "C" {

int32_t fastegnn_comm_unique_id_bytes(void) { return (int32_t)sizeof(nccl_uid_t); }

int fastegnn_comm_unique_id(void *id) {
  FE_REQUIRE(id, "fastegnn_comm_unique_id: null pointer");
  int rc = rccl_ready();
  if (rc) return rc;
  return rccl_check(g_rccl.GetUniqueId(static_cast<nccl_uid_t *>(id)), "ncclGetUniqueId");
}

int fastegnn_comm_init(fastegnn_comm_t **out, const void *id, int32_t rank, int32_t world) {
  FE_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, "fastegnn_comm_init: bad arguments");
  int rc = rccl_ready();
  if (rc) return rc;
  nccl_uid_t uid = *static_cast<const nccl_uid_t *>(id);
  nccl_comm_t c = nullptr;
  if ((rc = rccl_check(g_rccl.CommInitRank(&c, world, uid, rank), "ncclCommInitRank"))) return rc;
  *out = new fastegnn_comm{c, rank, world};
  return FASTEGNN_OK;
}

int fastegnn_comm_destroy(fastegnn_comm_t *c) {
  if (!c) return FASTEGNN_OK;
  int rc = rccl_check(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}

int32_t fastegnn_comm_rank(const fastegnn_comm_t *c) { return c ? c->rank : -1; }
int32_t fastegnn_comm_world(const fastegnn_comm_t *c) { return c ? c->world : 0; }

// buf[i] <- sum over ranks of buf[i]   (xsum, pools, g_pools, parameter gradients)
int fastegnn_comm_all_reduce(fastegnn_comm_t *c, float *buf, size_t n, void *stream) {
  FE_REQUIRE(c && (buf || n == 0), "fastegnn_comm_all_reduce: null pointer");
  if (n == 0) return FASTEGNN_OK;
  return rccl_check(g_rccl.AllReduce(buf, buf, n, NCCL_FLOAT32, NCCL_SUM, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

// out[r * n .. (r+1) * n) <- rank r's in[0 .. n)   (the padded source table)
int fastegnn_comm_all_gather(fastegnn_comm_t *c, const float *in, float *out, size_t n_per_rank, void *stream) {
  FE_REQUIRE(c && in && out, "fastegnn_comm_all_gather: null pointer");
  return rccl_check(g_rccl.AllGather(in, out, n_per_rank, NCCL_FLOAT32, c->comm, (hipStream_t)stream), "ncclAllGather");
}

// out[0 .. n) <- sum over ranks r' of rank r''s in[rank * n .. (rank+1) * n)   (transpose of the all-gather)
int fastegnn_comm_reduce_scatter(fastegnn_comm_t *c, const float *in, float *out, size_t n_per_rank, void *stream) {
  FE_REQUIRE(c && in && out, "fastegnn_comm_reduce_scatter: null pointer");
  return rccl_check(g_rccl.ReduceScatter(in, out, n_per_rank, NCCL_FLOAT32, NCCL_SUM, c->comm, (hipStream_t)stream),
                    "ncclReduceScatter");
}

// Rows of `row_floats` floats: rank r receives send_rows[r] rows from this rank (taken from `send` in rank order) and
// this rank receives recv_rows[r] rows from rank r (written to `recv` in rank order): the halo exchange and its
// transpose.  One grouped set of point-to-point transfers -- on the xGMI mesh every pair has its own link.
int fastegnn_comm_all_to_all_v(fastegnn_comm_t *c, const float *send, const int64_t *send_rows, float *recv,
                               const int64_t *recv_rows, int32_t row_floats, void *stream) {
  FE_REQUIRE(c && send_rows && recv_rows && row_floats > 0, "fastegnn_comm_all_to_all_v: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc = rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
  if (rc) return rc;
  size_t so = 0, ro = 0;
  for (int r = 0; r < c->world && !rc; ++r) {
    const size_t ns = (size_t)send_rows[r] * row_floats, nr = (size_t)recv_rows[r] * row_floats;
    if (r == c->rank) {   // own share: a device copy on the same stream
      if (ns != nr) { fe::set_error("fastegnn_comm_all_to_all_v: own send/recv counts differ"); rc = FASTEGNN_E_INVALID; break; }
      if (ns && hipMemcpyAsync(recv + ro, send + so, ns * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        fe::set_error("fastegnn_comm_all_to_all_v: local copy failed");
        rc = FASTEGNN_E_LAUNCH;
      }
    } else {
      if (ns) rc = rccl_check(g_rccl.Send(send + so, ns, NCCL_FLOAT32, r, c->comm, st), "ncclSend");
      if (!rc && nr) rc = rccl_check(g_rccl.Recv(recv + ro, nr, NCCL_FLOAT32, r, c->comm, st), "ncclRecv");
    }
    so += ns;
    ro += nr;
  }
  int rc2 = rccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
  return rc ? rc : rc2;
}

// ---- ghost-row pack / unpack of the halo exchange (rows of `w` floats, w % 4 == 0) ----
}  // extern "C"

namespace fe {
__global__ void gather_rows_kernel(const float *table, const int64_t *ids, long n, int w4, float *out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * w4) return;
  const long r = i / w4;
  const int k = (int)(i % w4);
  reinterpret_cast<f32x4 *>(out)[i] = reinterpret_cast<const f32x4 *>(table)[ids[r] * w4 + k];
}
// table[ids[r]] += rows[r]; an id may occur once per peer, so several rows can hit one table row: float atomics
__global__ void scatter_add_rows_kernel(float *table, const int64_t *ids, long n, int w, const float *rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * w) return;
  atomicAdd(&table[ids[i / w] * w + i % w], rows[i]);
}
// *flag = 1 when a value of a / b is Inf or NaN (the exponent field is all ones); grid-stride, one store per offending wave.
// A plain store of a constant (every writer writes the same value): `flag` may be a host-mapped word, and a store crosses PCIe on
// every platform where a device atomic on host memory may not
__global__ __launch_bounds__(256) void check_finite_kernel(const unsigned *a, long na, const unsigned *b, long nb, volatile int *flag) {
  bool bad = false;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) bad |= (a[i] & 0x7f800000u) == 0x7f800000u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) bad |= (b[i] & 0x7f800000u) == 0x7f800000u;
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) {
    *flag = 1;
    __threadfence_system();
  }
}
// buf = 0 when the word is set (one read of the word per workgroup)
__global__ __launch_bounds__(256) void zero_if_flagged_kernel(f32x4 *buf, long n4, float *tail, int ntail, const volatile int *flag) {
  __shared__ int set;
  if (threadIdx.x == 0) set = *flag;
  __syncthreads();
  if (!set) return;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) buf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
}  // namespace fe

extern "C" {
int fastegnn_f16_operands(void) { return (FE_FWD_F16 | FE_BWD_F16) != 0 ? 1 : 0; }
int fastegnn_check_finite(const float *a, int64_t na, const float *b, int64_t nb, int32_t *flag, void *stream) {
  FE_REQUIRE(flag && (a || na == 0) && (b || nb == 0) && na >= 0 && nb >= 0, "fastegnn_check_finite: null pointer");
  if (na + nb == 0) return FASTEGNN_OK;
  long n = na > nb ? na : nb;
  int grid = fe::cdiv(n, 256 * 8);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(fe::check_finite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned *>(a),
                     (long)na, reinterpret_cast<const unsigned *>(b), (long)nb, flag);
  return fe::check_launch("check_finite_kernel");
}
int fastegnn_host_words_alloc(int32_t n, int32_t **words) {
  FE_REQUIRE(words && n > 0, "fastegnn_host_words_alloc: bad arguments");
  void *p = nullptr;
  if (hipHostMalloc(&p, (size_t)n * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess || !p) {
    (void)hipGetLastError();
    fe::set_error("fastegnn_host_words_alloc: hipHostMalloc failed");
    return FASTEGNN_E_LAUNCH;
  }
  memset(p, 0, (size_t)n * sizeof(int32_t));
  *words = static_cast<int32_t *>(p);
  return FASTEGNN_OK;
}
int fastegnn_host_words_free(int32_t *words) {
  if (words && hipHostFree(words) != hipSuccess) {
    (void)hipGetLastError();
    fe::set_error("fastegnn_host_words_free: hipHostFree failed");
    return FASTEGNN_E_LAUNCH;
  }
  return FASTEGNN_OK;
}
int fastegnn_zero_if_flagged(float *buf, int64_t n, const int32_t *flag, void *stream) {
  FE_REQUIRE(flag && (buf || n == 0) && n >= 0, "fastegnn_zero_if_flagged: null pointer");
  FE_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "fastegnn_zero_if_flagged: buf must be 16-byte aligned");
  if (n == 0) return FASTEGNN_OK;
  const long n4 = n / 4;
  int grid = fe::cdiv(n4 > 0 ? n4 : 1, 256 * 4);
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(fe::zero_if_flagged_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<fe::f32x4 *>(buf), n4,
                     buf + 4 * n4, (int)(n - 4 * n4), flag);
  return fe::check_launch("zero_if_flagged_kernel");
}
int fastegnn_gather_rows(const float *table, const int64_t *ids, int64_t n, int32_t width, float *out, void *stream) {
  FE_REQUIRE(width > 0 && width % 4 == 0, "fastegnn_gather_rows: width must be a multiple of 4");
  if (n == 0) return FASTEGNN_OK;
  FE_REQUIRE(table && ids && out, "fastegnn_gather_rows: null pointer");
  hipLaunchKernelGGL(fe::gather_rows_kernel, dim3(fe::cdiv(n * (width / 4), 256)), dim3(256), 0, (hipStream_t)stream, table, ids,
                     (long)n, width / 4, out);
  return fe::check_launch("gather_rows_kernel");
}
int fastegnn_scatter_add_rows(float *table, const int64_t *ids, int64_t n, int32_t width, const float *rows, void *stream) {
  if (n == 0) return FASTEGNN_OK;
  FE_REQUIRE(table && ids && rows && width > 0, "fastegnn_scatter_add_rows: null pointer");
  hipLaunchKernelGGL(fe::scatter_add_rows_kernel, dim3(fe::cdiv(n * width, 256)), dim3(256), 0, (hipStream_t)stream, table, ids,
                     (long)n, width, rows);
  return fe::check_launch("scatter_add_rows_kernel");
}
}  // extern "C"
