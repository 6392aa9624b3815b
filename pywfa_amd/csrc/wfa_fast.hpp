// wfa_fast.hpp — what the register-resident kernels share (wfa_seg.hpp, wfa_lane.hpp, wfa_band.hpp): the launch
// arguments of the short-read kernels, the staging limits, and the DPP moves that fetch the k-1 / k+1 neighbour of a
// diagonal from the adjacent lane.  (Round 1's one- and two-alignments-per-wave kernels lived here; the segmented
// kernels superseded them and they were removed in round 2.)
#pragma once
#include "wfa_rtc_compat.hpp"
#include "wfa_common.hpp"

namespace wfa {

#define WFA_FAST_MAX_LEN 512
#define WFA_FAST_WORDS (WFA_FAST_MAX_LEN / 16 + 2)

// value of lane-1 (lane 0 receives `fill`) / lane+1 (lane 63 receives `fill`): gfx9 wave-shift DPP
__device__ __forceinline__ int from_lane_below(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int from_lane_above(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

struct FastArgs {
  const uint32_t* words;
  const WfaPairMeta* meta;
  const uint32_t* worklist;  // nullptr = identity
  const uint32_t* nwork_dev; // non-null: count read from device memory (leftovers of a previous stage)
  uint32_t nwork;
  int32_t* score;
  int32_t* status;
  uint32_t* fb_list;
  uint32_t* fb_count;
  int g;  // score step = gcd(x, o+e, e)
  // full-CIGAR variant of the segmented kernel (wfa_seg.hpp): history slot per work item of this launch
  int32_t* hist;          // slot t: hist + t * hist_stride: piggy-back code records of W bytes (one per step), then the walk's events and runs
  long long hist_stride;  // ints per slot
  int4* end_state;        // per slot {end score, end k, end offset, 1 = walk it}
  uint32_t work_begin;    // first work item of this launch (slot = item - work_begin)
  // full-CIGAR form of the lane kernel (wfa_lane_kernel<.., FULL>): the origin codes of every wave-step, 64 lanes x 8 bytes per record
  uint2* codes;           // wave w owns the records [w * codes_cap, (w + 1) * codes_cap)
  int codes_cap;          // records per wave
  // the general score-only form of the lane kernel (wfa_lane_kernel<.., HEUR>): free ends, wf-adaptive, step limit
  int ef, pbf, pef, tbf, tef;                                // ends-free span with these free ends (R/wavefront_termination.c:115-162)
  int heur, min_wf_len, max_dist_thr, steps_between;         // 1 = wf-adaptive (R/wavefront_heuristic.c:257-293)
  int max_steps;                                             // INT_MAX = unlimited (R/wavefront_unialign.c:98-107)
  int xdrop;                                                 // heur = 2: X-drop (R/wavefront_heuristic.c:297-383), the segmented form only
};

// neighbour diagonals inside a segment of W lanes (wfa_seg.hpp): lanes at a segment border receive NULL
template <int W>
__device__ __forceinline__ int seg_from_below(int v) {
  if (W <= 16) {
    int r = __builtin_amdgcn_update_dpp(WFA_OFFSET_NULL, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    if (W == 8 && (threadIdx.x & 7) == 0) r = WFA_OFFSET_NULL;
    return r;
  }
  int r = __builtin_amdgcn_update_dpp(WFA_OFFSET_NULL, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  if (W == 32 && (threadIdx.x & 31) == 0) r = WFA_OFFSET_NULL;
  return r;
}
template <int W>
__device__ __forceinline__ int seg_from_above(int v) {
  if (W <= 16) {
    int r = __builtin_amdgcn_update_dpp(WFA_OFFSET_NULL, v, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
    if (W == 8 && (threadIdx.x & 7) == 7) r = WFA_OFFSET_NULL;
    return r;
  }
  int r = __builtin_amdgcn_update_dpp(WFA_OFFSET_NULL, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  if (W == 32 && (threadIdx.x & 31) == 31) r = WFA_OFFSET_NULL;
  return r;
}

static inline int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

}  // namespace wfa
