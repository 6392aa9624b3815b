// k_biwfa.hip — translation units of the BiWFA kernel (wfa_biwfa.hpp), one per component count
// (-DWFA_TU_INDEX = 0 / 1 / 2 for NCOMP = 1 / 3 / 5, csrc/build.sh).
#include "wfa_biwfa.hpp"

namespace wfa {
#if WFA_TU_INDEX == 0
int launch_biwfa_c1(bool packed, const BiwfaArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_biwfa_ncomp<1>(packed, a, grid, smem, stream); }
#elif WFA_TU_INDEX == 1
int launch_biwfa_c3(bool packed, const BiwfaArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_biwfa_ncomp<3>(packed, a, grid, smem, stream); }
#elif WFA_TU_INDEX == 2
int launch_biwfa_c5(bool packed, const BiwfaArgs& a, int grid, size_t smem, hipStream_t stream) { return launch_biwfa_ncomp<5>(packed, a, grid, smem, stream); }
#else
#error "WFA_TU_INDEX: 0, 1, 2 for NCOMP = 1, 3, 5"
#endif
}  // namespace wfa
