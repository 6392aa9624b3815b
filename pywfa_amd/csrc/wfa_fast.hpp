// wfa_fast.hpp — register-resident short-read kernel (the C2 hot loop): one alignment per 64-lane
// workgroup, lane <-> diagonal k = lane - 32, the whole M/I/D wavefront history that compute-next
// needs held in VGPRs, the 2-bit packed pattern/text staged once in LDS, neighbour diagonals k-1/k+1
// fetched with wave-shift DPP moves, trimming / termination / window checks done with wave ballots.
//
// Scope: gap-affine, match = 0, no heuristic, score only, end-to-end (or ends-free with all free
// ends 0, which terminates on the same cell), both sequences <= 512 bases.  Exactly the reference's
// recurrences (R/wavefront_compute_affine.c:44-86) and end-trimming (R/wavefront_compute.c:571-605):
//   * an offset outside a wavefront's trimmed [lo,hi] reads as NULL: here every lane outside holds NULL;
//   * only M is clamped when out of bounds; interior out-of-bounds I/D values are kept;
//   * all scores are multiples of g = gcd(x, o+e, e), so the loop steps the score by g.
// A pair whose wavefront touches the edge of the 64-diagonal window (or is too long) is appended to
// the fallback list and finished by the general kernel (wfa_general.hpp) — results are identical
// because both kernels compute the same wavefronts.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
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
  int32_t* hist;          // slot t: hist + t * hist_stride, records of W entries {M, I, D, -} x int16
  long long hist_stride;  // ints per slot
  int4* end_state;        // per slot {end score, end k, end offset, 1 = walk it}
  uint32_t work_begin;    // first work item of this launch (slot = item - work_begin)
};

// X, OE, E: mismatch, gap_open+gap_extend, gap_extend in units of g.
// Work assignment: a wave takes CHUNKS of 64 consecutive pairs — one coalesced load of the 64 metadata
// records, lane j keeps the result of pair j, one coalesced store of 64 scores / statuses at the end —
// so every HBM line of the batch is touched by exactly one wave (no sector over-fetch across XCDs,
// no single-dword result stores); the next pair's packed words are prefetched into registers while
// the current pair is aligned.
template <int X, int OE, int E>
__global__ void __launch_bounds__(64)
wfa_fast_kernel(const FastArgs a) {
  constexpr int DM = (X > OE) ? X : OE;  // depth of the M history
  __shared__ uint32_t sP[WFA_FAST_WORDS];
  __shared__ uint32_t sT[WFA_FAST_WORDS];
  const int lane = threadIdx.x;
  const int k = lane - 32;
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  const uint32_t nchunks = (nwork + 63u) >> 6;

  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint32_t base = chunk << 6;
    const int cnt = (int)min(64u, nwork - base);
    const uint32_t my_pair = (lane < cnt) ? (a.worklist ? a.worklist[base + lane] : base + lane) : 0u;
    WfaPairMeta my_meta = a.meta[my_pair];
    int my_score = 0;
    bool my_fb = false;
    // prefetch the first pair's words: lanes 0..31 pattern word `lane`, lanes 32..63 text word `lane-32`
    uint32_t next_w = 0;
    {
      const uint32_t woff = (lane < 32) ? __builtin_amdgcn_readlane(my_meta.p_woff, 0) : __builtin_amdgcn_readlane(my_meta.t_woff, 0);
      const int len = (lane < 32) ? __builtin_amdgcn_readlane(my_meta.plen, 0) : __builtin_amdgcn_readlane(my_meta.tlen, 0);
      const int idx = lane & 31;
      if (len <= WFA_FAST_MAX_LEN && idx < ((len + 15) >> 4)) next_w = a.words[woff + idx];
    }
    for (int j = 0; j < cnt; ++j) {
      const int plen = __builtin_amdgcn_readlane(my_meta.plen, j);
      const int tlen = __builtin_amdgcn_readlane(my_meta.tlen, j);
      const int ak = tlen - plen;
      bool fallback = (plen > WFA_FAST_MAX_LEN) || (tlen > WFA_FAST_MAX_LEN) || (ak < -30) || (ak > 29);
      // stage this pair's words (prefetched), then prefetch the next pair's
      __syncthreads();
      if (lane < 32) sP[lane] = next_w; else sT[lane - 32] = next_w;
      if (lane == 0) { sP[32] = 0u; sP[33] = 0u; sT[32] = 0u; sT[33] = 0u; }
      next_w = 0;
      if (j + 1 < cnt) {
        const uint32_t woff = (lane < 32) ? __builtin_amdgcn_readlane(my_meta.p_woff, j + 1) : __builtin_amdgcn_readlane(my_meta.t_woff, j + 1);
        const int len = (lane < 32) ? __builtin_amdgcn_readlane(my_meta.plen, j + 1) : __builtin_amdgcn_readlane(my_meta.tlen, j + 1);
        const int idx = lane & 31;
        if (len <= WFA_FAST_MAX_LEN && idx < ((len + 15) >> 4)) next_w = a.words[woff + idx];
      }
      __syncthreads();
      int result = 0;
      if (!fallback) {
        int Mh[DM], Ih[E], Dh[E];
#pragma unroll
        for (int q = 0; q < DM; ++q) Mh[q] = WFA_OFFSET_NULL;
#pragma unroll
        for (int q = 0; q < E; ++q) { Ih[q] = WFA_OFFSET_NULL; Dh[q] = WFA_OFFSET_NULL; }
        int cur = (k == 0) ? 0 : WFA_OFFSET_NULL;  // wavefront 0 (R/wavefront_aligner.c:251-310)
        const int lim = min(tlen, plen + k);
        int edge = -1;  // AND of every offset this lane has held: non-negative once one was live
        int step = 0;
        for (;;) {
          // ---------------- extend M[s] (R/wavefront_extend_kernels.c:64-110) ----------------
          // in-bounds <=> offset <= lim, lim = min(tlen, plen + k); remaining run length = lim - offset
          {
            const bool live = cur >= 0;
            int h = max(cur, 0), v = max(cur - k, 0);
            int left = live ? lim - cur : 0;  // dead lanes: nothing left to compare
            if (__any(left > 0)) {
              bool more;
              do {
                // 32 bases per iteration: three packed words per sequence, two funnel shifts each
                const int pi = v >> 4, ti = h >> 4;
                const uint32_t p0 = sP[pi], p1 = sP[pi + 1], p2 = sP[pi + 2];
                const uint32_t t0 = sT[ti], t1 = sT[ti + 1], t2 = sT[ti + 2];
                const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
                const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
                int m = xl ? (__builtin_ctz(xl) >> 1) : (xh ? 16 + (__builtin_ctz(xh) >> 1) : 32);
                m = min(m, left);
                v += m; h += m; left -= m;
                more = (m == 32) && (left > 0);
              } while (__any(more));
              cur = live ? h : cur;
            }
          }
          // ---------------- termination (R/wavefront_termination.c:37-61) ----------------
          if (__builtin_amdgcn_readlane(cur, ak + 32) >= tlen) { result = -(step * a.g); break; }
          // ---------------- compute-next for score s+g ----------------
#pragma unroll
          for (int q = DM - 1; q > 0; --q) Mh[q] = Mh[q - 1];
          Mh[0] = cur;
          const int mx = Mh[X - 1], mo = Mh[OE - 1], ie = Ih[E - 1], de = Dh[E - 1];
          int ni = WFA_OFFSET_NULL, nd = WFA_OFFSET_NULL, nm = WFA_OFFSET_NULL;
          if (__any((mx & mo & ie & de) >= 0)) {  // some input offset is not NULL-ish
            ni = max(from_lane_below(mo, WFA_OFFSET_NULL), from_lane_below(ie, WFA_OFFSET_NULL)) + 1;
            nd = max(from_lane_above(mo, WFA_OFFSET_NULL), from_lane_above(de, WFA_OFFSET_NULL));
            nm = max(nd, max(mx + 1, ni));
            if (nm > lim) nm = WFA_OFFSET_NULL;  // only M is clamped (negative values are dead already)
            // ends of I and D (R/wavefront_compute.c:571-605).  Trimming only changes anything when a
            // LIVE offset is out of bounds (dead lanes are NULL-ish already): rare, near the sequence ends
            if (__any(max(ni, nd) > lim)) {
              const unsigned long long bi = __ballot(ni >= 0 && ni <= lim), bd = __ballot(nd >= 0 && nd <= lim);
              const int ilo = bi ? (int)__builtin_ctzll(bi) : 64, ihi = bi ? 63 - (int)__builtin_clzll(bi) : -1;
              const int dlo = bd ? (int)__builtin_ctzll(bd) : 64, dhi = bd ? 63 - (int)__builtin_clzll(bd) : -1;
              if (lane < ilo || lane > ihi) ni = WFA_OFFSET_NULL;
              if (lane < dlo || lane > dhi) nd = WFA_OFFSET_NULL;
            }
            edge &= nm & ni & nd;
          }
#pragma unroll
          for (int q = E - 1; q > 0; --q) { Ih[q] = Ih[q - 1]; Dh[q] = Dh[q - 1]; }
          Ih[0] = ni; Dh[0] = nd;
          cur = nm;
          ++step;
          // a live diagonal on either edge lane may spill out of the 64-diagonal window
          if ((__ballot(edge >= 0) & 0x8000000000000001ull) || step >= 4096) { fallback = true; break; }
        }
      }
      if (lane == j) { my_score = result; my_fb = fallback; }
    }
    // ---- coalesced results; compacted append of the leftovers ----
    if (lane < cnt) {
      if (!my_fb) a.score[my_pair] = my_score;
      a.status[my_pair] = my_fb ? WFA_INTERNAL_FALLBACK : 0;
    }
    const unsigned long long fbm = __ballot(my_fb && lane < cnt);
    if (fbm) {
      uint32_t slot = 0;
      if (lane == 0) slot = atomicAdd(a.fb_count, (uint32_t)__builtin_popcountll(fbm));
      slot = __builtin_amdgcn_readfirstlane(slot);
      if (my_fb && lane < cnt) a.fb_list[slot + __builtin_popcountll(fbm & ((1ull << lane) - 1ull))] = my_pair;
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Two alignments per wave: lanes 0-31 hold the 32 diagonals k = -16..15 of pair A, lanes 32-63 those of
// pair B (most 150 bp / 2 % wavefronts are < 20 diagonals wide, so a 64-diagonal window leaves 3/4 of
// the lanes idle).  Same recurrences as wfa_fast_kernel; per-pair quantities (lengths, in-bounds limit,
// LDS base of the staged words) are per-lane registers selected by the half.  The wave-shift DPP moves
// cross the half boundary, but only ever carry a dead (negative) value across it: a pair whose wavefront
// becomes live on one of ITS edge lanes (0/31 or 32/63) is handed to the next stage at once and all its
// registers are set to NULL, and so are those of a pair that has finished.
#define WFA_FAST2_MAX_LEN 240
#define WFA_FAST2_WORDS 18   // 16 words (256 bases) + 2 pad words per staged sequence

template <int X, int OE, int E>
__global__ void __launch_bounds__(64)
wfa_fast2_kernel(const FastArgs a) {
  constexpr int DM = (X > OE) ? X : OE;
  __shared__ uint32_t sW[4 * WFA_FAST2_WORDS];  // P_A, T_A, P_B, T_B
  const int lane = threadIdx.x;
  const int half = lane >> 5;
  const int k = (lane & 31) - 16;
  const int pidx0 = half * 2 * WFA_FAST2_WORDS;          // staged pattern of this half
  const int tidx0 = pidx0 + WFA_FAST2_WORDS;             // staged text of this half
  const int slot = lane >> 4, widx = lane & 15;          // staging role: sequence `slot`, word `widx`
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  const uint32_t nchunks = (nwork + 63u) >> 6;

  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint32_t base = chunk << 6;
    const int cnt = (int)min(64u, nwork - base);
    const uint32_t my_pair = (lane < cnt) ? (a.worklist ? a.worklist[base + lane] : base + lane) : 0u;
    WfaPairMeta my_meta = a.meta[my_pair];
    if (lane >= cnt) { my_meta.plen = 0; my_meta.tlen = 0; }
    int my_score = 0;
    bool my_fb = false;
    // word this lane stages for the couple (j, j+1): sequence `slot` of the couple, word `widx`
    auto fetch_word = [&](int j) -> uint32_t {
      const int ja = j, jb = min(j + 1, 63);
      const uint32_t woff = (slot == 0) ? __builtin_amdgcn_readlane(my_meta.p_woff, ja)
                          : (slot == 1) ? __builtin_amdgcn_readlane(my_meta.t_woff, ja)
                          : (slot == 2) ? __builtin_amdgcn_readlane(my_meta.p_woff, jb)
                                        : __builtin_amdgcn_readlane(my_meta.t_woff, jb);
      const int len = (slot == 0) ? __builtin_amdgcn_readlane(my_meta.plen, ja)
                    : (slot == 1) ? __builtin_amdgcn_readlane(my_meta.tlen, ja)
                    : (slot == 2) ? __builtin_amdgcn_readlane(my_meta.plen, jb)
                                  : __builtin_amdgcn_readlane(my_meta.tlen, jb);
      uint32_t w = 0;
      if (len <= WFA_FAST2_MAX_LEN && widx < ((len + 15) >> 4) && (slot < 2 || j + 1 < cnt)) w = a.words[woff + widx];
      return w;
    };
    uint32_t next_w = fetch_word(0);
    for (int j = 0; j < cnt; j += 2) {
      const bool has_b = (j + 1 < cnt);
      const int plen_a = __builtin_amdgcn_readlane(my_meta.plen, j), tlen_a = __builtin_amdgcn_readlane(my_meta.tlen, j);
      const int plen_b = __builtin_amdgcn_readlane(my_meta.plen, min(j + 1, 63)), tlen_b = __builtin_amdgcn_readlane(my_meta.tlen, min(j + 1, 63));
      const int ak_a = tlen_a - plen_a, ak_b = tlen_b - plen_b;
      bool fb_a = (plen_a > WFA_FAST2_MAX_LEN) || (tlen_a > WFA_FAST2_MAX_LEN) || (ak_a < -14) || (ak_a > 13);
      bool fb_b = has_b && ((plen_b > WFA_FAST2_MAX_LEN) || (tlen_b > WFA_FAST2_MAX_LEN) || (ak_b < -14) || (ak_b > 13));
      bool done_a = fb_a, done_b = fb_b || !has_b;
      int res_a = 0, res_b = 0;
      __syncthreads();
      sW[slot * WFA_FAST2_WORDS + widx] = next_w;
      if (lane < 8) sW[(lane >> 1) * WFA_FAST2_WORDS + 16 + (lane & 1)] = 0u;
      next_w = (j + 2 < cnt) ? fetch_word(j + 2) : 0u;
      __syncthreads();
      const int plen = half ? plen_b : plen_a, tlen = half ? tlen_b : tlen_a;
      const int lim = min(tlen, plen + k);
      int Mh[DM], Ih[E], Dh[E];
#pragma unroll
      for (int q = 0; q < DM; ++q) Mh[q] = WFA_OFFSET_NULL;
#pragma unroll
      for (int q = 0; q < E; ++q) { Ih[q] = WFA_OFFSET_NULL; Dh[q] = WFA_OFFSET_NULL; }
      const bool dead0 = half ? done_b : done_a;
      int cur = (k == 0 && !dead0) ? 0 : WFA_OFFSET_NULL;
      int edge = -1;
      int step = 0;
      // set every register of one half to NULL (that pair has finished or was handed on)
      auto kill = [&](int h) {
        if (half == h) {
          cur = WFA_OFFSET_NULL; edge = -1;
#pragma unroll
          for (int q = 0; q < DM; ++q) Mh[q] = WFA_OFFSET_NULL;
#pragma unroll
          for (int q = 0; q < E; ++q) { Ih[q] = WFA_OFFSET_NULL; Dh[q] = WFA_OFFSET_NULL; }
        }
      };
      while (!(done_a && done_b)) {
        // ---------------- extend ----------------
        {
          const bool live = cur >= 0;
          int h = max(cur, 0), v = max(cur - k, 0);
          int left = live ? lim - cur : 0;
          if (__any(left > 0)) {
            bool more;
            do {
              const int pi = pidx0 + (v >> 4), ti = tidx0 + (h >> 4);
              const uint32_t p0 = sW[pi], p1 = sW[pi + 1], p2 = sW[pi + 2];
              const uint32_t t0 = sW[ti], t1 = sW[ti + 1], t2 = sW[ti + 2];
              const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
              const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
              int m = xl ? (__builtin_ctz(xl) >> 1) : (xh ? 16 + (__builtin_ctz(xh) >> 1) : 32);
              m = min(m, left);
              v += m; h += m; left -= m;
              more = (m == 32) && (left > 0);
            } while (__any(more));
            cur = live ? h : cur;
          }
        }
        // ---------------- termination, per pair ----------------
        if (!done_a && __builtin_amdgcn_readlane(cur, ak_a + 16) >= tlen_a) { done_a = true; res_a = -(step * a.g); kill(0); }
        if (!done_b && __builtin_amdgcn_readlane(cur, ak_b + 48) >= tlen_b) { done_b = true; res_b = -(step * a.g); kill(1); }
        if (done_a && done_b) break;
        // ---------------- compute-next ----------------
#pragma unroll
        for (int q = DM - 1; q > 0; --q) Mh[q] = Mh[q - 1];
        Mh[0] = cur;
        const int mx = Mh[X - 1], mo = Mh[OE - 1], ie = Ih[E - 1], de = Dh[E - 1];
        int ni = WFA_OFFSET_NULL, nd = WFA_OFFSET_NULL, nm = WFA_OFFSET_NULL;
        if (__any((mx & mo & ie & de) >= 0)) {
          ni = max(from_lane_below(mo, WFA_OFFSET_NULL), from_lane_below(ie, WFA_OFFSET_NULL)) + 1;
          nd = max(from_lane_above(mo, WFA_OFFSET_NULL), from_lane_above(de, WFA_OFFSET_NULL));
          nm = max(nd, max(mx + 1, ni));
          if (nm > lim) nm = WFA_OFFSET_NULL;
          if (__any(max(ni, nd) > lim)) {
            // trim the ends of I and D inside each half (R/wavefront_compute.c:571-605)
            const unsigned long long bi = __ballot(ni >= 0 && ni <= lim), bd = __ballot(nd >= 0 && nd <= lim);
            const uint32_t bih = half ? (uint32_t)(bi >> 32) : (uint32_t)bi, bdh = half ? (uint32_t)(bd >> 32) : (uint32_t)bd;
            const int hl = lane & 31;
            const int ilo = bih ? __builtin_ctz(bih) : 32, ihi = bih ? 31 - __builtin_clz(bih) : -1;
            const int dlo = bdh ? __builtin_ctz(bdh) : 32, dhi = bdh ? 31 - __builtin_clz(bdh) : -1;
            if (hl < ilo || hl > ihi) ni = WFA_OFFSET_NULL;
            if (hl < dlo || hl > dhi) nd = WFA_OFFSET_NULL;
          }
          edge &= nm & ni & nd;
        }
#pragma unroll
        for (int q = E - 1; q > 0; --q) { Ih[q] = Ih[q - 1]; Dh[q] = Dh[q - 1]; }
        Ih[0] = ni; Dh[0] = nd;
        cur = nm;
        ++step;
        const unsigned long long eb = __ballot(edge >= 0);
        if (!done_a && ((eb & 0x0000000080000001ull) || step >= 4096)) { done_a = true; fb_a = true; kill(0); }
        if (!done_b && ((eb & 0x8000000100000000ull) || step >= 4096)) { done_b = true; fb_b = true; kill(1); }
      }
      if (lane == j) { my_score = res_a; my_fb = fb_a; }
      if (lane == j + 1) { my_score = res_b; my_fb = fb_b; }
    }
    if (lane < cnt) {
      if (!my_fb) a.score[my_pair] = my_score;
      a.status[my_pair] = my_fb ? WFA_INTERNAL_FALLBACK : 0;
    }
    const unsigned long long fbm = __ballot(my_fb && lane < cnt);
    if (fbm) {
      uint32_t slot_ = 0;
      if (lane == 0) slot_ = atomicAdd(a.fb_count, (uint32_t)__builtin_popcountll(fbm));
      slot_ = __builtin_amdgcn_readfirstlane(slot_);
      if (my_fb && lane < cnt) a.fb_list[slot_ + __builtin_popcountll(fbm & ((1ull << lane) - 1ull))] = my_pair;
    }
  }
}

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

// which configurations the fast kernel covers
inline bool fast_supported(const WfaDevConfig& c, int ncomp, bool full) {
  if (full || ncomp != 3 || c.match != 0 || c.heuristic != 0 || c.wildcard >= 0) return false;
  if (c.endsfree && (c.pbf | c.pef | c.tbf | c.tef)) return false;
  if (c.max_steps != INT_MAX) return false;
  const int g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  const int X = c.x / g, OE = (c.o1 + c.e1) / g, E = c.e1 / g;
  return (X == 2 && OE == 4 && E == 1);  // pywfa's default penalties 4/6/2 (and multiples)
}

inline int launch_fast(const WfaDevConfig& c, int cu_count, hipStream_t stream, const uint32_t* words,
                       const WfaPairMeta* meta, const uint32_t* worklist, const uint32_t* nwork_dev, uint32_t nwork,
                       int32_t* score, int32_t* status, uint32_t* fb_list, uint32_t* fb_count, int variant) {
  // variant: 0 = one alignment per wave, 1 = two per wave in half-waves
  FastArgs a;
  a.words = words; a.meta = meta; a.worklist = worklist; a.nwork_dev = nwork_dev; a.nwork = nwork;
  a.score = score; a.status = status; a.fb_list = fb_list; a.fb_count = fb_count;
  a.g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  a.hist = nullptr; a.hist_stride = 0; a.end_state = nullptr; a.work_begin = 0;
  const char* env = getenv("WFA_HIP_FAST_WAVES_PER_CU");
  const int per_cu = (env && *env) ? atoi(env) : 32;
  long long grid = (long long)cu_count * per_cu;
  const long long nchunks = ((long long)nwork + 63) / 64;
  if (grid > nchunks) grid = nchunks;
  if (nwork_dev) grid = (long long)cu_count * 32;  // leftovers: count known only on the device
  if (grid < 1) grid = 1;
  if (variant == 1) hipLaunchKernelGGL((wfa_fast2_kernel<2, 4, 1>), dim3((unsigned)grid), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((wfa_fast_kernel<2, 4, 1>), dim3((unsigned)grid), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace wfa
