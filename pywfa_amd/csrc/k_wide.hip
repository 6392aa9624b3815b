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
template <bool FULL, bool TWO>
static int launch_wide_ft(bool grows, bool w32, const WideArgs& a, int grid, int threads, size_t smem, hipStream_t stream) {
  if (grows) return w32 ? launch_wide_t<FULL, TWO, true, true>(a, grid, threads, smem, stream) : launch_wide_t<FULL, TWO, true, false>(a, grid, threads, smem, stream);
  return w32 ? launch_wide_t<FULL, TWO, false, true>(a, grid, threads, smem, stream) : launch_wide_t<FULL, TWO, false, false>(a, grid, threads, smem, stream);
}
// grows: the rows live in the workgroup's slice of the workspace (a.rows), otherwise in LDS; w32: int32 offsets (reads beyond 16 kb)
int launch_wide(bool full, bool two, bool grows, const WideArgs& a, int grid, int threads, size_t smem, hipStream_t stream, bool w32) {
  if (two) return full ? launch_wide_ft<true, true>(grows, w32, a, grid, threads, smem, stream) : launch_wide_ft<false, true>(grows, w32, a, grid, threads, smem, stream);
  return full ? launch_wide_ft<true, false>(grows, w32, a, grid, threads, smem, stream) : launch_wide_ft<false, false>(grows, w32, a, grid, threads, smem, stream);
}
}  // namespace wfa
