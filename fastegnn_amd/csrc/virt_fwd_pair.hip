// The two-waves-per-tile instantiation of virt_fwd_kernel (virt_fwd.h), in a translation unit of its own: compiled WITHOUT the
// max-memory-clause scheduler strategy of the other stage kernels (Makefile) -- this hipcc's backend crashes on it with that option.
#include "virt_fwd.h"

namespace fe {

void launch_virt_fwd_pair(const VirtArgs &a, int grid, size_t lds, hipStream_t st) {
  if constexpr (GM_VIRT_FWD == GM_F16)
    hipLaunchKernelGGL((virt_fwd_kernel<GM_F16, true>), dim3(grid), dim3(64 * VIRT_WAVES), lds, st, a);
  else
    (void)a, (void)grid, (void)lds, (void)st;   // (the wide-range build never takes the PAIR walk: virt_forward's `pair` is false there)
}

}  // namespace fe
