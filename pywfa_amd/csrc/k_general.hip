// k_general.hip — translation unit of the general kernel (wfa_general.hpp) for ONE component count, compiled once
// per index (-DWFA_TU_INDEX = 0 / 1 / 2 for NCOMP = 1 / 3 / 5, csrc/build.sh).
#include "wfa_general.hpp"

namespace wfa {
#if WFA_TU_INDEX == 0
int launch_general_c1(bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream) {
  return launch_general_ncomp<1>(packed, full, pb, a, grid, threads, stream);
}
#elif WFA_TU_INDEX == 1
int launch_general_c3(bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream) {
  return launch_general_ncomp<3>(packed, full, pb, a, grid, threads, stream);
}
#elif WFA_TU_INDEX == 2
int launch_general_c5(bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream) {
  return launch_general_ncomp<5>(packed, full, pb, a, grid, threads, stream);
}
#else
#error "WFA_TU_INDEX: 0, 1, 2 for NCOMP = 1, 3, 5"
#endif
}  // namespace wfa
