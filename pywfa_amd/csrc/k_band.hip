// k_band.hip — translation unit of the banded register kernels (wfa_band.hpp) for ONE penalty shape, compiled once per
// index (-DWFA_TU_INDEX=i, csrc/build.sh): 0..3 = the gap-affine shapes of WFA_BAND_SHAPES, 4 = gap-affine-2p
// 4/6/2/24/1, 5 = the walks over the history and the expansion of their run records into op bytes.
#if WFA_TU_INDEX == 5
#define WFA_BAND_WALK_KERNELS 1
#endif
#include "wfa_band.hpp"

namespace wfa {
#define WFA_BAND_DEFINE(i, x, oe, e, oe2, e2)                                                                           \
  int launch_band_s##i(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds, long long grid, hipStream_t s) { \
    return launch_band_shape<x, oe, e, oe2, e2>(a, nch, full, adapt, seqlds, grid, s);                                  \
  }
#if WFA_TU_INDEX == 0
WFA_BAND_DEFINE(0, 2, 4, 1, 0, 0)
#elif WFA_TU_INDEX == 1
WFA_BAND_DEFINE(1, 2, 3, 1, 0, 0)
#elif WFA_TU_INDEX == 2
WFA_BAND_DEFINE(2, 4, 7, 1, 0, 0)
#elif WFA_TU_INDEX == 3
WFA_BAND_DEFINE(3, 3, 5, 1, 0, 0)
#elif WFA_TU_INDEX == 4
WFA_BAND_DEFINE(4, 4, 8, 2, 25, 1)
#elif WFA_TU_INDEX == 5
int launch_band_bt(const BandArgs& a, int nch, hipStream_t stream) { return launch_band_bt_impl(a, nch, stream); }
int launch_lane_expand(const BandArgs& a, hipStream_t walk_stream, hipStream_t expand_stream) { return launch_lane_expand_impl(a, walk_stream, expand_stream); }
#else
#error "WFA_TU_INDEX: 0..3 gap-affine shapes, 4 gap-affine-2p, 5 walks"
#endif
}  // namespace wfa
