// k_seg.hip — translation unit of the segmented register kernels (wfa_seg.hpp) for ONE penalty shape:
// compiled once per shape index (-DWFA_TU_INDEX=i, csrc/build.sh) so that the shapes build in parallel.
#include "wfa_seg.hpp"

namespace wfa {
#define WFA_SEG_DEFINE(i, x, oe, e)                                                                              \
  int launch_seg_s##i(int w, bool lazy, unsigned grid, hipStream_t stream, const FastArgs& a) {                  \
    return launch_seg_shape<x, oe, e>(w, lazy, grid, stream, a);                                                 \
  }                                                                                                              \
  int launch_seg_full_s##i(int w, unsigned grid, hipStream_t stream, const FastArgs& a) {                        \
    return launch_seg_full_shape<x, oe, e>(w, grid, stream, a);                                                  \
  }                                                                                                              \
  int launch_seg_heur_s##i(unsigned grid, hipStream_t stream, const FastArgs& a) {                               \
    return launch_seg_heur_shape<x, oe, e>(grid, stream, a);                                                     \
  }
#if WFA_TU_INDEX == 0
WFA_SEG_DEFINE(0, 2, 4, 1)
#elif WFA_TU_INDEX == 1
WFA_SEG_DEFINE(1, 2, 3, 1)
#elif WFA_TU_INDEX == 2
WFA_SEG_DEFINE(2, 4, 7, 1)
#elif WFA_TU_INDEX == 3
WFA_SEG_DEFINE(3, 3, 5, 1)
#elif WFA_TU_INDEX == 4
WFA_SEG_DEFINE(4, 6, 8, 3)
#elif WFA_TU_INDEX == 5
WFA_SEG_DEFINE(5, 5, 3, 3)
#elif WFA_TU_INDEX == 6
WFA_SEG_DEFINE(6, 1, 2, 1)
#else
#error "WFA_TU_INDEX must name a shape of WFA_SEG_SHAPES"
#endif
}  // namespace wfa
