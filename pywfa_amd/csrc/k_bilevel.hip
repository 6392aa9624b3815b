// k_bilevel.hip — translation units of the level-by-level BiWFA kernels (wfa_bilevel.hpp), one per component count
// (-DWFA_TU_INDEX = 0 / 1 / 2 for NCOMP = 1 / 3 / 5; the seed and finish kernels live in unit 0; csrc/build.sh).
#include "wfa_bilevel.hpp"

namespace wfa {
#if WFA_TU_INDEX == 0
int launch_bl_split_c1(bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_split_ncomp<1>(packed, i16, threads, seql, a, grid, smem, stream); }
int launch_bl_base_c1(bool packed, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_base_ncomp<1>(packed, a, grid, smem, stream); }
int launch_bl_split_lds_c1(int threads, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_split_lds_ncomp<1>(threads, a, grid, smem, stream); }
int launch_bl_seed(const BlArgs& a, hipStream_t stream) {
  if (a.k.nwork == 0) return 0;
  hipLaunchKernelGGL(bl_seed_kernel<0>, dim3((a.k.nwork + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_bl_finish(const BlArgs& a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(bl_finish_kernel<0>, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#elif WFA_TU_INDEX == 1
int launch_bl_split_c3(bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_split_ncomp<3>(packed, i16, threads, seql, a, grid, smem, stream); }
int launch_bl_base_c3(bool packed, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_base_ncomp<3>(packed, a, grid, smem, stream); }
int launch_bl_split_lds_c3(int threads, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_split_lds_ncomp<3>(threads, a, grid, smem, stream); }
#elif WFA_TU_INDEX == 2
int launch_bl_split_c5(bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_split_ncomp<5>(packed, i16, threads, seql, a, grid, smem, stream); }
int launch_bl_base_c5(bool packed, const BlArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_bl_base_ncomp<5>(packed, a, grid, smem, stream); }
#else
#error "WFA_TU_INDEX: 0, 1, 2 for NCOMP = 1, 3, 5"
#endif
}  // namespace wfa
