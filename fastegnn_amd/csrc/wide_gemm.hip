// Host side of the wide path's two GEMM kernels (wide_gemm.h): one instance per (column quadrants, fused activation mode, prefetch
// depth).  The instances take six minutes to compile and depend on none of the build variants' switches, so this file is compiled
// ONCE per part (-DFE_WG_PART=0..5: the Makefile's wide_gemm_p*.o, in parallel) and linked into every library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef FE_ACT_GENERIC
#define FE_ACT_GENERIC
#endif
#define FE_WIDE_GEMM_IMPL
#include "wide_gemm.h"

#ifndef FE_WG_PART
#error "compile with -DFE_WG_PART=0..5"
#endif

namespace fe {
namespace wide {

#define FE_X3_ONE(NQ_, P_, E_, D_) hipLaunchKernelGGL((gemm_x3_kernel<NQ_, P_, E_, D_>), grid, dim3(XWAVES * 64), x3_lds_bytes(NQ_) + extra, st, g)
// the seven modes every shape has: plain, activation prologue (SiLU / any), activation-backward epilogue, head-output epilogue
#define FE_X3_MODES(NQ_, D_)                                               \
  do {                                                                     \
    if (pm == AM_SILU) FE_X3_ONE(NQ_, AM_SILU, AM_NONE, D_);               \
    else if (pm == AM_GEN) FE_X3_ONE(NQ_, AM_GEN, AM_NONE, D_);            \
    else if (em == AM_SILU) FE_X3_ONE(NQ_, AM_NONE, AM_SILU, D_);          \
    else if (em == AM_GEN) FE_X3_ONE(NQ_, AM_NONE, AM_GEN, D_);            \
    else if (em == AM_DOT_SILU) FE_X3_ONE(NQ_, AM_NONE, AM_DOT_SILU, D_);  \
    else if (em == AM_DOT_GEN) FE_X3_ONE(NQ_, AM_NONE, AM_DOT_GEN, D_);    \
    else FE_X3_ONE(NQ_, AM_NONE, AM_NONE, D_);                             \
  } while (0)

#if FE_WG_PART == 0
void launch_gemm_x3_nq12(const GemmX3 &g, int nq, int pm, int em, dim3 grid, hipStream_t st) {
  const int extra = 0;
  if (nq == 1) FE_X3_MODES(1, false);
  else FE_X3_MODES(2, false);
}
#elif FE_WG_PART == 1
void launch_gemm_x3_nq3(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st) {
  const int extra = 0;
  FE_X3_MODES(3, false);
}
#elif FE_WG_PART == 2
void launch_gemm_x3_nq4(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st) {
  const int extra = 0;
  FE_X3_MODES(4, false);
}
#elif FE_WG_PART == 3
void launch_gemm_x3_deep(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st) {
  const int extra = 0;
  if (pm == AM_SILU) FE_X3_ONE(4, AM_SILU, AM_NONE, true);
  else if (pm == AM_GEN) FE_X3_ONE(4, AM_GEN, AM_NONE, true);
  else if (em == AM_SILU) FE_X3_ONE(4, AM_NONE, AM_SILU, true);
  else if (em == AM_GEN) FE_X3_ONE(4, AM_NONE, AM_GEN, true);
  else FE_X3_ONE(4, AM_NONE, AM_NONE, true);
}
#elif FE_WG_PART == 4
// the head forms: the output epilogue, and the generated-operand prologue -- which exists in the four-buffer form only: in the
// one-buffer form the compiler keeps the stream's state in scratch memory (1 KB per lane, 25 x the time)
void launch_gemm_x3_deep_head(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st) {
  const int extra = pm >= AM_HEAD_SILU ? ((g.Kd + 127) / 128) * 128 * (int)sizeof(float) : 0;   // w2 behind the strips
  if (pm == AM_HEAD_SILU) FE_X3_ONE(4, AM_HEAD_SILU, AM_NONE, true);
  else if (pm == AM_HEAD_GEN) FE_X3_ONE(4, AM_HEAD_GEN, AM_NONE, true);
  else if (em == AM_DOT_SILU) FE_X3_ONE(4, AM_NONE, AM_DOT_SILU, true);
  else FE_X3_ONE(4, AM_NONE, AM_DOT_GEN, true);
}
#else
void launch_gemm_x3_nq12(const GemmX3 &g, int nq, int pm, int em, dim3 grid, hipStream_t st);
void launch_gemm_x3_nq3(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st);
void launch_gemm_x3_nq4(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st);
void launch_gemm_x3_deep(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st);
void launch_gemm_x3_deep_head(const GemmX3 &g, int pm, int em, dim3 grid, hipStream_t st);

int launch_gemm_x3(const GemmX3 &g, int nq, int pm, int em, bool deep, dim3 grid, hipStream_t st) {
  if (pm >= AM_HEAD_SILU && !(deep && nq == 4)) return 1;   // (the host asks for the generated-operand form on DEEP shapes only)
  if (deep && nq == 4) {
    if (pm >= AM_HEAD_SILU || em >= AM_DOT_SILU) launch_gemm_x3_deep_head(g, pm, em, grid, st);
    else launch_gemm_x3_deep(g, pm, em, grid, st);
  } else if (nq <= 2) launch_gemm_x3_nq12(g, nq, pm, em, grid, st);
  else if (nq == 3) launch_gemm_x3_nq3(g, pm, em, grid, st);
  else launch_gemm_x3_nq4(g, pm, em, grid, st);
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
#endif

}  // namespace wide
}  // namespace fe
