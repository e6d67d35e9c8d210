// Host side of the wide path's two GEMM kernels (wide_gemm.h): one instance per (column quadrants, fused activation mode, prefetch
// depth).  Its own translation unit -- the instances take two minutes to compile and depend on none of the build variants' switches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef FE_ACT_GENERIC
#define FE_ACT_GENERIC
#endif
#define FE_WIDE_GEMM_IMPL
#include "wide_gemm.h"

namespace fe {
namespace wide {

int launch_gemm_x3(const GemmX3 &g, int nq, int pm, int em, bool deep, dim3 grid, hipStream_t st) {
  const dim3 block(XWAVES * 64);
  // the generated-operand (head) prologue exists in the four-buffer form only: in the one-buffer form the compiler keeps the
  // stream's state in scratch memory (1 KB per lane, 25 x the time) -- the host asks for it on DEEP shapes only
  if (pm >= AM_HEAD_SILU && !(deep && nq == 4)) return 1;
  const int extra = pm >= AM_HEAD_SILU ? ((g.Kd + 127) / 128) * 128 * (int)sizeof(float) : 0;   // w2 behind the strips
#define FE_X3_LAUNCH(NQ_, P_, E_)                                                                                          \
  do {                                                                                                                     \
    if (NQ_ == 4 && deep) hipLaunchKernelGGL((gemm_x3_kernel<4, P_, E_, true>), grid, block, x3_lds_bytes(4) + extra, st, g); \
    else hipLaunchKernelGGL((gemm_x3_kernel<NQ_, P_, E_, false>), grid, block, x3_lds_bytes(NQ_) + extra, st, g);          \
  } while (0)
#define FE_X3_MODES(NQ_)                                         \
  do {                                                           \
    if (pm == AM_SILU) FE_X3_LAUNCH(NQ_, AM_SILU, AM_NONE);      \
    else if (pm == AM_GEN) FE_X3_LAUNCH(NQ_, AM_GEN, AM_NONE);   \
    else if (pm == AM_HEAD_SILU) FE_X3_LAUNCH(4, AM_HEAD_SILU, AM_NONE); \
    else if (pm == AM_HEAD_GEN) FE_X3_LAUNCH(4, AM_HEAD_GEN, AM_NONE);   \
    else if (em == AM_SILU) FE_X3_LAUNCH(NQ_, AM_NONE, AM_SILU); \
    else if (em == AM_GEN) FE_X3_LAUNCH(NQ_, AM_NONE, AM_GEN);   \
    else if (em == AM_DOT_SILU) FE_X3_LAUNCH(NQ_, AM_NONE, AM_DOT_SILU); \
    else if (em == AM_DOT_GEN) FE_X3_LAUNCH(NQ_, AM_NONE, AM_DOT_GEN);   \
    else FE_X3_LAUNCH(NQ_, AM_NONE, AM_NONE);                    \
  } while (0)
  switch (nq) {
    case 1: FE_X3_MODES(1); break;
    case 2: FE_X3_MODES(2); break;
    case 3: FE_X3_MODES(3); break;
    default: FE_X3_MODES(4); break;
  }
#undef FE_X3_MODES
#undef FE_X3_LAUNCH
  return 0;
}

int launch_tn_x3(const TnX3 &t, int pm, int gm, dim3 grid, hipStream_t st) {
  if (gm == AM_SILU) {   // (the generated-gradient forms take X as it is, or as a SiLU pre-activation)
    if (pm == AM_SILU) hipLaunchKernelGGL((tn_x3_kernel<AM_SILU, AM_SILU>), grid, dim3(256), 0, st, t);
    else if (pm == AM_GEN) hipLaunchKernelGGL((tn_x3_kernel<AM_GEN, AM_SILU>), grid, dim3(256), 0, st, t);
    else hipLaunchKernelGGL((tn_x3_kernel<AM_NONE, AM_SILU>), grid, dim3(256), 0, st, t);
    return 0;
  }
  if (gm == AM_GEN) {
    if (pm != AM_NONE) hipLaunchKernelGGL((tn_x3_kernel<AM_GEN, AM_GEN>), grid, dim3(256), 0, st, t);
    else hipLaunchKernelGGL((tn_x3_kernel<AM_NONE, AM_GEN>), grid, dim3(256), 0, st, t);
    return 0;
  }
  switch (pm) {
    case AM_SILU: hipLaunchKernelGGL((tn_x3_kernel<AM_SILU, AM_NONE>), grid, dim3(256), 0, st, t); break;
    case AM_GEN: hipLaunchKernelGGL((tn_x3_kernel<AM_GEN, AM_NONE>), grid, dim3(256), 0, st, t); break;
    default: hipLaunchKernelGGL((tn_x3_kernel<AM_NONE, AM_NONE>), grid, dim3(256), 0, st, t); break;
  }
  return 0;
}

}  // namespace wide
}  // namespace fe
