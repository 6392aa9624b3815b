// k_wide.hip — translation unit of the wide-wavefront kernel (wfa_wide.hpp).
#include "wfa_wide.hpp"

namespace wfa {
template <bool FULL, bool TWO, bool GROWS, bool W32 = false>
static int launch_wide_t(const WideArgs& a, int grid, int threads, size_t smem, hipStream_t stream) {
  // (more than the default 64 KB of dynamic LDS: a workgroup may take the CU's whole 160 KB)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_wide_kernel<FULL, TWO, GROWS, W32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  hipLaunchKernelGGL((wfa_wide_kernel<FULL, TWO, GROWS, W32>), dim3(grid), dim3(threads), smem, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_wide(bool full, bool two, const WideArgs& a, int grid, int threads, size_t smem, hipStream_t stream, bool w32) {
  if (w32) {   // int32 rows in the workspace: reads beyond 16 kb
    if (two) return full ? launch_wide_t<true, true, true, true>(a, grid, threads, smem, stream) : launch_wide_t<false, true, true, true>(a, grid, threads, smem, stream);
    return full ? launch_wide_t<true, false, true, true>(a, grid, threads, smem, stream) : launch_wide_t<false, false, true, true>(a, grid, threads, smem, stream);
  }
  if (!two && a.rows) return full ? launch_wide_t<true, false, true>(a, grid, threads, smem, stream) : launch_wide_t<false, false, true>(a, grid, threads, smem, stream);
  if (two) return full ? launch_wide_t<true, true, true>(a, grid, threads, smem, stream) : launch_wide_t<false, true, true>(a, grid, threads, smem, stream);
  return full ? launch_wide_t<true, false, false>(a, grid, threads, smem, stream) : launch_wide_t<false, false, false>(a, grid, threads, smem, stream);
}
}  // namespace wfa
