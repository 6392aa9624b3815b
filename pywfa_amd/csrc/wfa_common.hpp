// wfa_common.hpp — types shared by the device kernels and the host side of libwfa_hip.so.
//
// Vocabulary follows the reference's domain (WFA2-lib, /root/reference/pywfa/WFA2_lib/wavefront):
//   diagonal k = h - v, offset = h (text position), v = offset - k (wavefront_offset.h:50-57);
//   a wavefront = {lo, hi, offsets[k]} per score and component M / I1 / D1 / I2 / D2
//   (wavefront.h:56-77); NULL offset = INT32_MIN/2 (wavefront_offset.h:44).
#pragma once
#include "wfa_rtc_compat.hpp"

#define WFA_OFFSET_NULL (-1073741824)  // INT32_MIN/2

// per-pair internal status written by kernels (never returned to callers)
#define WFA_INTERNAL_FALLBACK  (-1000)  // fast kernel: pair does not fit its window -> general kernel
#define WFA_INTERNAL_OVERFLOW  (-1001)  // general kernel: wavefront arena too small -> retry with a larger one

// Per-pair metadata resident in HBM (16 B): word offsets of the 2-bit packed sequences and lengths.
struct WfaPairMeta {
  uint32_t p_woff;  // first u32 word of the packed pattern
  uint32_t t_woff;  // first u32 word of the packed text
  int32_t plen;
  int32_t tlen;
};

// Penalties after wavefront_penalties_set_affine/affine2p (wavefront_penalties.c:95-173) and the
// rest of the per-aligner configuration, in the form the kernels consume.
struct WfaDevConfig {
  int32_t metric; // WFA_DIST_* (0 indel, 1 edit, 2 linear, 3 affine, 4 affine2p)
  int32_t match;  // <= 0 (original match score)
  int32_t x, o1, e1, o2, e2;  // adjusted penalties
  int32_t scope;  // max_score_scope (wavefront_components.c:81-124)
  int32_t endsfree;
  int32_t pbf, pef, tbf, tef;  // pattern/text begin/end free
  int32_t heuristic;           // WFA_HEUR_*
  int32_t min_wf_len, max_dist_thr, steps_between, xdrop;
  int32_t max_steps;           // INT32_MAX = unlimited
  int32_t wildcard;            // -1 none
  int32_t score_mode;          // how a completed pair's score is reported from the kernels' -s (csrc/wfa_hip.hip derive_dev_config):
                               // 0 as it is; 1 (sw_match (plen + tlen) - s) / 2 (match < 0, R/wavefront_penalties.h:73); 2 +s (indel / edit)
  int32_t sw_match;            // -match of the original configuration (score_mode 1)
  int32_t rtc;                 // 1: penalty shapes without an instantiation are compiled at run time (csrc/wfa_rtc.cpp)
  int32_t lin;                 // 1 (host side): a one-component distance with CIGARs mapped onto the gap-affine register kernels' LIN form
                               // (wfa_lane.hpp); the general kernel keeps the original one-component configuration
  int32_t biwfa_top;           // 1: the general kernel stands in for the top-level base case of BiWFA (reads of <= 100 bases whose
                               // score outgrows the BiWFA kernel's base-case history): a completed pair keeps the unset score
                               // (SURVEY Q6), every other ending is "unattainable" (R/wavefront_bialign.c:182-187,725-729)
};

// Arguments of the alignment kernels (passed by value).
struct WfaKernelArgs {
  // sequences
  const uint32_t* words;    // 2-bit packed, 16 bases per u32, base i of a sequence in bits [2i,2i+2)
  const uint8_t* bytes;     // ASCII blob (8-bit path)
  const WfaPairMeta* meta;
  const int64_t* p_boff;    // byte offsets (8-bit path)
  const int64_t* t_boff;
  // work list: pair ids to process (nullptr = identity); count read from *nwork_dev when non-null
  const uint32_t* worklist;
  const uint32_t* nwork_dev;
  const uint32_t* wbeg_dev;   // non-null: first list position of this launch, read from device memory (see BandArgs::wbeg_dev)
  uint32_t nwork;
  // results
  int32_t* score;
  int32_t* status;
  uint8_t* cigar_ops;
  const int64_t* cigar_off;
  int64_t* cigar_begin;
  int32_t* cigar_len;
  // per-workgroup wavefront workspace (general kernel)
  int32_t* ws;
  int64_t ws_stride;  // int32 elements per workgroup
  // fallback list produced by the fast kernel
  uint32_t* fb_list;
  uint32_t* fb_count;
  WfaDevConfig cfg;
};

// ---- what the register-resident kernels share (wfa_seg.hpp, wfa_lane.hpp, wfa_band.hpp, wfa_slim.hpp): the launch arguments of the
// short-read kernels, the staging limits, and the DPP moves that fetch the k-1 / k+1 neighbour of a diagonal from the adjacent lane
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
  // the lane kernel's score-only forms: slices of the work list taken at run time (round 5).  Non-null: a device counter (zero at launch);
  // a wave takes `dyn_chunk` pairs at a time from it until the list is used up, instead of one fixed slice per wave — the lanes of a
  // wave drain (few busy lanes, full instruction cost) once per kernel, not once per slice
  uint32_t* dyn_next;
  uint32_t dyn_chunk;
  int lin;                // 1: the one-component form of the FULL kernels (wfa_lane.hpp: LIN) — host side: selects the run-time instantiation
  int scope;              // heur = 2: max_score_scope (R/wavefront_components.c:81-124): more null steps than this end the alignment "unreachable"
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
