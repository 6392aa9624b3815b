// k_lane.hip — translation units of the lane-per-pair kernel (wfa_lane.hpp), one per penalty shape of WFA_SEG_SHAPES
// (-DWFA_TU_INDEX=i, csrc/build.sh).
#include "wfa_lane.hpp"

namespace wfa {
#define WFA_LANE_DEFINE(i, x, oe, e)                                                                                        \
  int launch_lane_s##i(unsigned grid, size_t smem, hipStream_t stream, const FastArgs& a, int slot_words, int refill_min, bool full, int heur) { \
    return launch_lane_shape<x, oe, e>(grid, smem, stream, a, slot_words, refill_min, full, heur);                                               \
  }
#if WFA_TU_INDEX == 0
WFA_LANE_DEFINE(0, 2, 4, 1)
#elif WFA_TU_INDEX == 1
WFA_LANE_DEFINE(1, 2, 3, 1)
#elif WFA_TU_INDEX == 2
WFA_LANE_DEFINE(2, 4, 7, 1)
#elif WFA_TU_INDEX == 3
WFA_LANE_DEFINE(3, 3, 5, 1)
#elif WFA_TU_INDEX == 4
WFA_LANE_DEFINE(4, 6, 8, 3)
#elif WFA_TU_INDEX == 5
WFA_LANE_DEFINE(5, 5, 3, 3)
#elif WFA_TU_INDEX == 6
WFA_LANE_DEFINE(6, 1, 2, 1)
#else
#error "WFA_TU_INDEX must name a shape of WFA_SEG_SHAPES"
#endif
}  // namespace wfa
