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
  uint32_t nwork;
  int32_t* score;
  int32_t* status;
  uint32_t* fb_list;
  uint32_t* fb_count;
  int g;  // score step = gcd(x, o+e, e)
};

// X, OE, E: mismatch, gap_open+gap_extend, gap_extend in units of g
template <int X, int OE, int E>
__global__ void __launch_bounds__(64)
wfa_fast_kernel(const FastArgs a) {
  constexpr int DM = (X > OE) ? X : OE;  // depth of the M history
  __shared__ uint32_t sP[WFA_FAST_WORDS];
  __shared__ uint32_t sT[WFA_FAST_WORDS];
  const int lane = threadIdx.x;
  const int k = lane - 32;

  for (uint32_t wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    const WfaPairMeta pm = a.meta[pair];
    const int plen = pm.plen, tlen = pm.tlen;
    const int ak = tlen - plen;
    bool fallback = (plen > WFA_FAST_MAX_LEN) || (tlen > WFA_FAST_MAX_LEN) || (ak < -30) || (ak > 29);
    int result = 0;
    if (!fallback) {
      // stage the packed sequences (one trailing word is read by the funnel shift)
      const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
      __syncthreads();  // previous pair's LDS reads are done
      if (lane <= nwp && lane < WFA_FAST_WORDS) sP[lane] = (lane < nwp) ? a.words[pm.p_woff + lane] : 0u;
      if (lane <= nwt && lane < WFA_FAST_WORDS) sT[lane] = (lane < nwt) ? a.words[pm.t_woff + lane] : 0u;
      __syncthreads();

      int Mh[DM], Ih[E], Dh[E];
#pragma unroll
      for (int j = 0; j < DM; ++j) Mh[j] = WFA_OFFSET_NULL;
#pragma unroll
      for (int j = 0; j < E; ++j) { Ih[j] = WFA_OFFSET_NULL; Dh[j] = WFA_OFFSET_NULL; }
      int cur = (k == 0) ? 0 : WFA_OFFSET_NULL;  // wavefront 0 (R/wavefront_aligner.c:251-310)
      int s = 0;
      bool done = false;
      for (int step = 0; step < 4096; ++step) {
        // ---------------- extend M[s] (R/wavefront_extend_kernels.c:64-110) ----------------
        if (__any(cur >= 0)) {
          bool active = cur >= 0;
          int h = cur, v = cur - k;
          int left = active ? min(plen - v, tlen - h) : 0;
          while (__any(active)) {
            const int vi = active ? v : 0, hi_ = active ? h : 0;
            const uint32_t p0 = sP[vi >> 4], p1 = sP[(vi >> 4) + 1];
            const uint32_t t0 = sT[hi_ >> 4], t1 = sT[(hi_ >> 4) + 1];
            const uint32_t pw = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)(vi & 15) << 1);
            const uint32_t tw = __builtin_amdgcn_alignbit(t1, t0, (uint32_t)(hi_ & 15) << 1);
            const uint32_t x = pw ^ tw;
            int m = x ? (__builtin_ctz(x) >> 1) : 16;
            m = min(m, left);
            if (active) { v += m; h += m; left -= m; }
            active = active && (m == 16) && (left > 0);
          }
          if (cur >= 0) cur = h;
          // ---------------- termination (R/wavefront_termination.c:37-61) ----------------
          const int at_end = __builtin_amdgcn_readlane(cur, ak + 32);
          if (at_end >= tlen) { done = true; result = -s; break; }
          // window check: a live diagonal on either edge lane may spill out of the 64-diagonal window
          const unsigned long long bm = __ballot(cur >= 0);
          if (bm & 0x8000000000000001ull) { fallback = true; break; }
        }
        // ---------------- compute-next for score s+g ----------------
        // history shift: Mh[0] = M[s], Mh[1] = M[s-g], ...
#pragma unroll
        for (int j = DM - 1; j > 0; --j) Mh[j] = Mh[j - 1];
        Mh[0] = cur;
        s += a.g;
        const int mx = Mh[X - 1], mo = Mh[OE - 1], ie = Ih[E - 1], de = Dh[E - 1];
        const bool any_in = __any((mx >= 0) | (mo >= 0) | (ie >= 0) | (de >= 0));
        int ni = WFA_OFFSET_NULL, nd = WFA_OFFSET_NULL, nm = WFA_OFFSET_NULL;
        if (any_in) {
          const int mo_lo = from_lane_below(mo, WFA_OFFSET_NULL), ie_lo = from_lane_below(ie, WFA_OFFSET_NULL);
          const int mo_hi = from_lane_above(mo, WFA_OFFSET_NULL), de_hi = from_lane_above(de, WFA_OFFSET_NULL);
          ni = max(mo_lo, ie_lo) + 1;
          nd = max(mo_hi, de_hi);
          nm = max(nd, max(mx + 1, ni));
          if ((uint32_t)nm > (uint32_t)tlen || (uint32_t)(nm - k) > (uint32_t)plen) nm = WFA_OFFSET_NULL;
          // trim the ends of I and D (R/wavefront_compute.c:571-605): outside [first,last] in-bounds -> NULL
          const bool inb_i = (uint32_t)ni <= (uint32_t)tlen && (uint32_t)(ni - k) <= (uint32_t)plen;
          const bool inb_d = (uint32_t)nd <= (uint32_t)tlen && (uint32_t)(nd - k) <= (uint32_t)plen;
          const unsigned long long bi = __ballot(inb_i), bd = __ballot(inb_d);
          const int ilo = bi ? (int)__builtin_ctzll(bi) : 64, ihi = bi ? 63 - (int)__builtin_clzll(bi) : -1;
          const int dlo = bd ? (int)__builtin_ctzll(bd) : 64, dhi = bd ? 63 - (int)__builtin_clzll(bd) : -1;
          if (lane < ilo || lane > ihi) ni = WFA_OFFSET_NULL;
          if (lane < dlo || lane > dhi) nd = WFA_OFFSET_NULL;
          if ((bi | bd) & 0x8000000000000001ull) { fallback = true; break; }
        }
#pragma unroll
        for (int j = E - 1; j > 0; --j) { Ih[j] = Ih[j - 1]; Dh[j] = Dh[j - 1]; }
        Ih[0] = ni; Dh[0] = nd;
        cur = nm;
      }
      if (!done) fallback = true;
    }
    if (lane == 0) {
      if (fallback) {
        a.status[pair] = WFA_INTERNAL_FALLBACK;
        a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;
      } else {
        a.score[pair] = result;
        a.status[pair] = 0;
      }
    }
  }
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
                       const WfaPairMeta* meta, const uint32_t* worklist, uint32_t nwork, int32_t* score,
                       int32_t* status, uint32_t* fb_list, uint32_t* fb_count) {
  FastArgs a;
  a.words = words; a.meta = meta; a.worklist = worklist; a.nwork = nwork;
  a.score = score; a.status = status; a.fb_list = fb_list; a.fb_count = fb_count;
  a.g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  const char* env = getenv("WFA_HIP_FAST_WAVES_PER_CU");
  const int per_cu = (env && *env) ? atoi(env) : 32;
  long long grid = (long long)cu_count * per_cu;
  if (grid > (long long)nwork) grid = nwork;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((wfa_fast_kernel<2, 4, 1>), dim3((unsigned)grid), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace wfa
