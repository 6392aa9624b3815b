// k_slim.hip — translation unit of the slim form of the banded kernel (wfa_slim.hpp) for ONE penalty shape, compiled once per
// index (-DWFA_TU_INDEX=i, csrc/build.sh): 0..3 = the gap-affine shapes of WFA_BAND_SHAPES, 4 = gap-affine-2p 4/6/2/24/1.
#include "wfa_slim.hpp"

namespace wfa {
#define WFA_SLIM_DEFINE(i, x, oe, e) \
  int launch_slim_s##i(const BandArgs& a, int nch, bool full, long long grid, hipStream_t s) { return launch_slim_shape<x, oe, e, 0, 0>(a, nch, full, grid, s); } \
  int launch_slim_mailbox_s##i(const BandArgs& a, bool full, hipStream_t s, SlimMailbox* mb) { return launch_slim_mailbox_shape<x, oe, e, 0, 0>(a, full, s, mb); }
#if WFA_TU_INDEX == 0
WFA_SLIM_DEFINE(0, 2, 4, 1)
#elif WFA_TU_INDEX == 1
WFA_SLIM_DEFINE(1, 2, 3, 1)
#elif WFA_TU_INDEX == 2
WFA_SLIM_DEFINE(2, 4, 7, 1)
#elif WFA_TU_INDEX == 3
WFA_SLIM_DEFINE(3, 3, 5, 1)
#elif WFA_TU_INDEX == 4
int launch_slim_s4(const BandArgs& a, int nch, bool full, long long grid, hipStream_t s) { return launch_slim_shape<4, 8, 2, 25, 1>(a, nch, full, grid, s); }
#else
#error "WFA_TU_INDEX: 0..3 = the gap-affine shapes of WFA_BAND_SHAPES, 4 = gap-affine-2p"
#endif
}  // namespace wfa
