// k_wide.hip — translation unit of the wide-wavefront kernel (wfa_wide.hpp).
#include "wfa_wide.hpp"

namespace wfa {
int launch_wide(bool full, const WideArgs& a, int grid, int threads, size_t smem, hipStream_t stream) {
  const void* kern = full ? reinterpret_cast<const void*>(&wfa_wide_kernel<true>) : reinterpret_cast<const void*>(&wfa_wide_kernel<false>);
  // (more than the default 64 KB of dynamic LDS: a workgroup may take the CU's whole 160 KB)
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) { (void)hipGetLastError(); return -1; }
  if (full) hipLaunchKernelGGL(wfa_wide_kernel<true>, dim3(grid), dim3(threads), smem, stream, a);
  else hipLaunchKernelGGL(wfa_wide_kernel<false>, dim3(grid), dim3(threads), smem, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace wfa
