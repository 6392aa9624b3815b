// wfa_biwfa.hpp — BiWFA on the device (memory_mode "biwfa" = the reference's wavefront_memory_ultralow, scope=full):
// R/wavefront_bialign.c restated for one WAVE per alignment.
//
// The reference keeps O(s) memory by never storing the wavefront history: a forward aligner (on the sequences) and a
// reverse aligner (on the reversed sequences), both score-only rings of `max_score_scope` wavefronts, advance in
// turns until their wavefronts overlap; the overlap gives a breakpoint {cell, component, score of each half}
// (R/wavefront_bialign.c:189-395, 411-519); the two halves are aligned recursively with the breakpoint's component
// as end / begin component (:581-658); a half whose score is <= 250 (or a window in which one direction reaches the
// end at a score <= 500 before any overlap) is aligned by the ordinary algorithm with a full history and a
// backtrace (:155-188, 520-548), and the op strings are concatenated left to right.
//
// Here: one 64-lane workgroup per pair, lanes over diagonals k, wave-synchronous (no inter-wave hand-offs); the two
// rings and the base-case history live in the workgroup's slice of the HBM workspace (L2-resident at these sizes),
// the per-score directory {lo, hi per component, data index} of each aligner in LDS; the recursion is an explicit
// stack in LDS.  Every decision the op string depends on is taken in the reference's order: forward / reverse turn
// order, the scan order of wavefront_bialign_overlap (scores of the other aligner newest first; D2, I2, D1, I1, M),
// the lowest diagonal among the overlapping ones, strict improvement of the breakpoint score, and the backtrace's
// candidate priority (R/wavefront_backtrace.c:49-59) in the base cases.
// Quirk kept: the score is written only after a top-level split (:651-656); a pair answered by
// the ordinary algorithm at the top level (both sequences <= 100 bases, or an end reached before any overlap) reports
// score INT32_MIN with status 0 and the right op string.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include "wfa_hip.h"
#include "wfa_common.hpp"
#include "wfa_general.hpp"

namespace wfa {

#define WFA_BI_FALLBACK_MIN_SCORE 250   // R/wavefront_bialign.c:48
#define WFA_BI_FALLBACK_MIN_LENGTH 100  // :49
#define WFA_BI_RECOVERY_MIN_SCORE 500   // :50
#define WFA_BI_BASE_SLOTS 512           // scores a base case can reach (<= 500, above) + slack
#define WFA_BI_STACK 96                 // pending windows (the recursion is depth-first: two per level)

// internal states of an aligner (R/wfa.h:52-55)
#define WFA_BI_OK (-1)
#define WFA_BI_END_REACHED (-2)
#define WFA_BI_END_UNREACHABLE (-3)

struct BiwfaArgs {
  WfaKernelArgs k;
  int64_t ring_ints;   // ints of ONE score-only ring (scope x NCOMP x ring stride)
  int64_t base_ints;   // ints of the base-case history
  int ring_stride;     // plen + tlen + 3 of the longest pair of the launch
  int base_stride;     // diagonals a base-case wavefront can span (2 x 501 + 3, or less for short reads)
  int score_only;      // scope=score with a step limit: the top-level breakpoint search alone (R/wavefront_bialign.c:662-702)
};

// a window of the two sequences, read forwards or backwards (R/wavefront_sequences.c:275-310)
template <bool PACKED>
struct BiView {
  const uint32_t* pw; const uint32_t* tw;   // packed words of the whole sequences
  const uint8_t* pb; const uint8_t* tb;     // bytes of the whole sequences (8-bit path)
  int pbeg, pend, tbeg, tend, wildcard;
  bool reverse;

  // length of the common prefix of pattern[v..] and text[h..] of the (possibly reversed) window, at most maxrun
  __device__ __forceinline__ int run(int v, int h, int maxrun) const {
    int n = 0;
    if (PACKED) {
      if (!reverse) {
        const int pv = pbeg + v, th = tbeg + h;
        while (n < maxrun) {
          const uint32_t x = window16(pw, pv + n) ^ window16(tw, th + n);
          const int m = x ? (__builtin_ctz(x) >> 1) : 16;
          n += m;
          if (m < 16) break;
        }
      } else {
        // base i of the reversed window is base (end - 1 - i) of the sequence: compare 16 bases downwards from q
        const int pq = pend - 1 - v, tq = tend - 1 - h;
        while (n < maxrun) {
          const int a = pq - n, b = tq - n;           // topmost base of this probe in each sequence
          // bases a-15 .. a in bits [0,32), base a in the top two bits; below position 0: shift the window up
          const int sa = (a >= 15) ? 0 : 15 - a, sb = (b >= 15) ? 0 : 15 - b;
          const uint32_t wa = window16(pw, a - 15 + sa) << (2 * sa), wb = window16(tw, b - 15 + sb) << (2 * sb);
          const uint32_t x = wa ^ wb;
          const int m = x ? (__builtin_clz(x) >> 1) : 16;
          n += m;
          if (m < 16) break;
        }
      }
      return min(n, maxrun);
    }
    while (n < maxrun) {
      const int pc = reverse ? pb[pend - 1 - v - n] : pb[pbeg + v + n];
      const int tc = reverse ? tb[tend - 1 - h - n] : tb[tbeg + h + n];
      if (!(pc == tc || (wildcard >= 0 && (pc == wildcard || tc == wildcard)))) break;
      ++n;
    }
    return n;
  }
};

// One unidirectional aligner of the wave: directory ring in LDS, offsets in HBM.
template <int NCOMP>
struct BiSide {
  typedef Meta<NCOMP> MT;
  int* ring;        // LDS: scope records of MT::INTS ints (score s -> slot s % scope)
  int* ws;          // HBM: offsets
  int* dir;         // HBM directory of every score (base case only; nullptr for the score-only rings)
  int stride;       // diagonals per component slot
  int rbase;        // diagonal of element 0
  int slots;        // data slots (scope for a ring, WFA_BI_BASE_SLOTS for the base case)
  int null_steps;
  int cur_lo, cur_hi, cur_idx0, cur_exists;   // M of the current score
  // heuristic state of a forward / reverse aligner (R/wavefront_heuristic.c:114-121: re-set at every breakpoint search)
  int steps_wait, have_max_sw, max_sw;

  __device__ __forceinline__ int data_index(int s) const { return (s % slots) * NCOMP * stride; }
};

// wavefront 0 with a begin component (R/wavefront_aligner.c:329-390): the cell (k = 0, offset 0) of that component
template <int NCOMP>
__device__ __forceinline__ void bi_side_init(BiSide<NCOMP>& sd, int scope, int comp_begin, int plen, int tlen, int lane) {
  typedef Meta<NCOMP> MT;
  // diagonals [rbase, rbase + stride): the whole matrix [-plen - 1, tlen + 1] when it fits, else centred on diagonal 0
  // (base case: scores <= 500 keep |k| <= 501)
  sd.rbase = (sd.stride >= plen + tlen + 3) ? -plen - 1 : -min(plen, (sd.stride - 3) / 2) - 1;
  sd.null_steps = 0;
  sd.steps_wait = 0; sd.have_max_sw = 0; sd.max_sw = 0;
  const int data = sd.data_index(0);
  __syncthreads();
  if (lane == 0) {
    int* m = sd.ring;
    for (int c = 0; c < NCOMP; ++c) { m[MT::LO + c] = 1; m[MT::HI + c] = -1; }
    m[MT::LO + comp_begin] = 0; m[MT::HI + comp_begin] = 0;
    m[MT::BASE] = sd.rbase; m[MT::WIDTH] = sd.stride; m[MT::DATA] = data; m[MT::EXISTS] = (comp_begin == 0) ? 1 : 0;
    sd.ws[data + comp_begin * sd.stride + (0 - sd.rbase)] = 0;
    if (sd.dir) for (int c = 0; c < MT::INTS; ++c) sd.dir[c] = m[c];
  }
  sd.cur_exists = (comp_begin == 0) ? 1 : 0;
  sd.cur_lo = sd.cur_exists ? 0 : 1; sd.cur_hi = sd.cur_exists ? 0 : -1;
  sd.cur_idx0 = data - sd.rbase;
  __syncthreads();
}

// R/wavefront_extend.c:90-125 / :178-214 without heuristic: extend M[s]; returns the largest antidiagonal 2*offset - k
template <int NCOMP, bool PACKED>
__device__ __forceinline__ int bi_side_extend(BiSide<NCOMP>& sd, const BiView<PACKED>& view, int plen, int tlen, int lane) {
  int best = 0;
  if (sd.cur_exists) {
    for (int k = sd.cur_lo + lane; k <= sd.cur_hi; k += 64) {
      const int off = sd.ws[sd.cur_idx0 + k];
      if (off == WFA_OFFSET_NULL) continue;
      const int h = off, v = off - k;
      const int ext = off + view.run(v, h, min(plen - v, tlen - h));
      if (ext != off) sd.ws[sd.cur_idx0 + k] = ext;
      best = max(best, 2 * ext - k);
    }
    best = wave_max(best);
  }
  __syncthreads();
  return best;
}

// Round 4: the heuristic cut-off of a forward / reverse aligner after its extension (R/wavefront_extend.c:117-123,206-212 ->
// R/wavefront_heuristic.c:509-567): wf-adaptive (:257-293) or X-drop (:297-383) on M[s], the gap wavefronts cut to the same limits
// (:161-172); one wave, the form of wfa_general_kernel's cut-off phase.  The base cases run without (R/wavefront_bialigner.c:66-68).
template <int NCOMP>
__device__ __forceinline__ void bi_side_cutoff(BiSide<NCOMP>& sd, const WfaDevConfig& cfg, int scope, int s, int plen, int tlen, int lane) {
  typedef Meta<NCOMP> MT;
  if (cfg.heuristic == 0 || !sd.cur_exists || sd.cur_lo > sd.cur_hi) return;
  --sd.steps_wait;
  const int cur_lo = sd.cur_lo, cur_hi = sd.cur_hi;
  int new_lo = cur_lo, new_hi = cur_hi;
  const int* ws = sd.ws;
  if (cfg.heuristic == 1) {
    if (sd.steps_wait <= 0 && (cur_hi - cur_lo + 1) >= cfg.min_wf_len) {
      int dmin = max(plen, tlen);
      for (int k = cur_lo + lane; k <= cur_hi; k += 64) {
        const int off = ws[sd.cur_idx0 + k];
        const int d = (off >= 0) ? max(plen - (off - k), tlen - off) : -WFA_OFFSET_NULL;
        dmin = min(dmin, d);
      }
      dmin = wave_min(dmin);
      int lc = INT_MAX, hc = INT_MIN;
      for (int k = cur_lo + lane; k <= cur_hi; k += 64) {
        const int off = ws[sd.cur_idx0 + k];
        const int d = (off >= 0) ? max(plen - (off - k), tlen - off) : -WFA_OFFSET_NULL;
        if (d - dmin <= cfg.max_dist_thr) { lc = min(lc, k); hc = max(hc, k); }
      }
      lc = wave_min(lc); hc = wave_max(hc);
      const int ak = tlen - plen;
      const int top_limit = min(ak, cur_hi);
      if (top_limit > cur_lo) new_lo = min(lc, top_limit);
      const int bottom_limit = max(ak, new_lo);
      if (bottom_limit < cur_hi) new_hi = max(hc, bottom_limit);
      sd.steps_wait = cfg.steps_between;
    }
  } else if (cfg.heuristic == 2) {
    if (sd.steps_wait <= 0) {
      const int g = (cfg.match != 0) ? -cfg.match : -1;  // R/wavefront_heuristic.c:306-307
      int cmax = INT_MIN, lc = INT_MAX, hc = INT_MIN;
      for (int k = cur_lo + lane; k <= cur_hi; k += 64) {
        const int off = ws[sd.cur_idx0 + k];
        if (off < 0) continue;
        const int sw = (g * ((off - k) + off) - s) / 2;
        cmax = max(cmax, sw);
        if (sd.have_max_sw && sd.max_sw - sw < cfg.xdrop) { lc = min(lc, k); hc = max(hc, k); }
      }
      cmax = wave_max(cmax); lc = wave_min(lc); hc = wave_max(hc);
      if (sd.have_max_sw) {
        if (lc == INT_MAX) { new_lo = cur_hi + 1; new_hi = cur_hi; }
        else { new_lo = lc; new_hi = hc; }
        if (cmax > sd.max_sw) sd.max_sw = cmax;
      } else {
        sd.max_sw = cmax; sd.have_max_sw = 1;
      }
      sd.steps_wait = cfg.steps_between;
    }
  }
  if (new_lo != cur_lo || new_hi != cur_hi) {
    sd.cur_lo = new_lo; sd.cur_hi = new_hi;
    __syncthreads();
    if (lane == 0) {
      int* m = sd.ring + (s % scope) * MT::INTS;
      m[MT::LO] = new_lo; m[MT::HI] = new_hi;
      for (int c = 1; c < NCOMP; ++c) {  // wf_heuristic_equate (R/wavefront_heuristic.c:161-172)
        if (m[MT::LO + c] <= m[MT::HI + c]) {
          m[MT::LO + c] = max(m[MT::LO + c], new_lo);
          m[MT::HI + c] = min(m[MT::HI + c], new_hi);
        }
      }
    }
    __syncthreads();
  }
}

// R/wavefront_termination.c:37-113: the end component's wavefront of score s holds offset >= tlen on diagonal tlen - plen.
// Only evaluated when M[s] exists (the test sits behind the `mwavefront == NULL` return of wavefront_extend_end2end).
template <int NCOMP>
__device__ __forceinline__ bool bi_side_terminated(const BiSide<NCOMP>& sd, int scope, int s, int comp_end, int plen, int tlen) {
  typedef Meta<NCOMP> MT;
  if (!sd.cur_exists) return false;
  const int* m = sd.ring + (s % scope) * MT::INTS;
  const int ak = tlen - plen;
  if (m[MT::LO + comp_end] > ak || ak > m[MT::HI + comp_end]) return false;
  return sd.ws[m[MT::DATA] + comp_end * m[MT::WIDTH] + (ak - m[MT::BASE])] >= tlen;
}

// compute-next for score s (R/wavefront_compute_affine.c:44-86,229-260, R/wavefront_compute_affine2p.c:45-106,334-368,
// R/wavefront_compute_edit.c / _linear.c for NCOMP = 1, limits R/wavefront_compute.c:40-86, trimming :571-605): the
// modular form of wfa_general_kernel, one wave.  Returns false when the data of score s would not fit (base case only).
template <int NCOMP>
__device__ __forceinline__ bool bi_side_compute(BiSide<NCOMP>& sd, const WfaDevConfig& cfg, int scope, int s, int plen, int tlen, int lane) {
  typedef Meta<NCOMP> MT;
  const int* ws = sd.ws;
  WfIn nullin; nullin.lo = 1; nullin.hi = -1; nullin.idx0 = 0;
  const WfIn mx = (NCOMP == 1 && cfg.metric == 0) ? nullin : fetch_in<NCOMP>(sd.ring, scope, s - cfg.x, 0);
  const WfIn mo1 = fetch_in<NCOMP>(sd.ring, scope, s - cfg.o1 - cfg.e1, 0);
  const WfIn i1e = (NCOMP == 1) ? nullin : fetch_in<NCOMP>(sd.ring, scope, s - cfg.e1, 1);
  const WfIn d1e = (NCOMP == 1) ? nullin : fetch_in<NCOMP>(sd.ring, scope, s - cfg.e1, 2);
  WfIn mo2 = nullin, i2e = nullin, d2e = nullin;
  if (NCOMP == 5) {
    mo2 = fetch_in<NCOMP>(sd.ring, scope, s - cfg.o2 - cfg.e2, 0);
    i2e = fetch_in<NCOMP>(sd.ring, scope, s - cfg.e2, 3);
    d2e = fetch_in<NCOMP>(sd.ring, scope, s - cfg.e2, 4);
  }
  const bool all_null = mx.null() && mo1.null() && i1e.null() && d1e.null() &&
                        (NCOMP != 5 || (mo2.null() && i2e.null() && d2e.null()));
  int* const mslot = sd.ring + (s % scope) * MT::INTS;
  bool ok = true;
  int tlo[NCOMP], thi[NCOMP];
  int base = 0, width = 0, data = 0, exists = 0;
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { tlo[c] = 1; thi[c] = -1; }
  if (all_null) {
    ++sd.null_steps;
    sd.cur_exists = 0; sd.cur_lo = 1; sd.cur_hi = -1; sd.cur_idx0 = 0;
  } else {
    sd.null_steps = 0;
    int lo = mx.lo, hi = mx.hi;
    lo = min(lo, mo1.lo - 1); hi = max(hi, mo1.hi + 1);
    if (NCOMP != 1) {
      lo = min(lo, i1e.lo + 1); hi = max(hi, i1e.hi + 1);
      lo = min(lo, d1e.lo - 1); hi = max(hi, d1e.hi - 1);
    }
    if (NCOMP == 5) {
      lo = min(lo, mo2.lo - 1); hi = max(hi, mo2.hi + 1);
      lo = min(lo, i2e.lo + 1); hi = max(hi, i2e.hi + 1);
      lo = min(lo, d2e.lo - 1); hi = max(hi, d2e.hi - 1);
    }
    const bool has_i1 = (NCOMP != 1) && (!mo1.null() || !i1e.null());
    const bool has_d1 = (NCOMP != 1) && (!mo1.null() || !d1e.null());
    const bool has_i2 = (NCOMP == 5) && (!mo2.null() || !i2e.null());
    const bool has_d2 = (NCOMP == 5) && (!mo2.null() || !d2e.null());
    base = sd.rbase; width = sd.stride; data = sd.data_index(s); exists = 1;
    if (lo < sd.rbase || hi >= sd.rbase + sd.stride) ok = false;   // (cannot happen for a ring sized plen + tlen + 3)
    if (ok) {
      const int o_m = data - base;
      const int o_i1 = o_m + width, o_d1 = o_m + 2 * width, o_i2 = o_m + 3 * width, o_d2 = o_m + 4 * width;
      int* wsw = sd.ws;
      int tmin[NCOMP], tmax[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { tmin[c] = INT_MAX; tmax[c] = INT_MIN; }
      for (int k = lo + lane; k <= hi; k += 64) {
        const int ins1 = max(mo1.get(ws, k - 1), i1e.get(ws, k - 1)) + 1;
        const int del1 = max(mo1.get(ws, k + 1), d1e.get(ws, k + 1));
        int ins = ins1, del = del1;
        if (has_i1) {
          wsw[o_i1 + k] = ins1;
          if ((uint32_t)ins1 <= (uint32_t)tlen && (uint32_t)(ins1 - k) <= (uint32_t)plen) { tmin[NCOMP > 1 ? 1 : 0] = min(tmin[NCOMP > 1 ? 1 : 0], k); tmax[NCOMP > 1 ? 1 : 0] = max(tmax[NCOMP > 1 ? 1 : 0], k); }
        }
        if (has_d1) {
          wsw[o_d1 + k] = del1;
          if ((uint32_t)del1 <= (uint32_t)tlen && (uint32_t)(del1 - k) <= (uint32_t)plen) { tmin[NCOMP > 2 ? 2 : 0] = min(tmin[NCOMP > 2 ? 2 : 0], k); tmax[NCOMP > 2 ? 2 : 0] = max(tmax[NCOMP > 2 ? 2 : 0], k); }
        }
        if (NCOMP == 5) {
          const int ins2 = max(mo2.get(ws, k - 1), i2e.get(ws, k - 1)) + 1;
          const int del2 = max(mo2.get(ws, k + 1), d2e.get(ws, k + 1));
          if (has_i2) {
            wsw[o_i2 + k] = ins2;
            if ((uint32_t)ins2 <= (uint32_t)tlen && (uint32_t)(ins2 - k) <= (uint32_t)plen) { tmin[NCOMP - 2] = min(tmin[NCOMP - 2], k); tmax[NCOMP - 2] = max(tmax[NCOMP - 2], k); }
          }
          if (has_d2) {
            wsw[o_d2 + k] = del2;
            if ((uint32_t)del2 <= (uint32_t)tlen && (uint32_t)(del2 - k) <= (uint32_t)plen) { tmin[NCOMP - 1] = min(tmin[NCOMP - 1], k); tmax[NCOMP - 1] = max(tmax[NCOMP - 1], k); }
          }
          ins = max(ins1, ins2);
          del = max(del1, del2);
        }
        int mv = (NCOMP == 1 && cfg.metric == 0) ? max(del, ins) : max(del, max(mx.get(ws, k) + 1, ins));
        if ((uint32_t)mv > (uint32_t)tlen || (uint32_t)(mv - k) > (uint32_t)plen) mv = WFA_OFFSET_NULL;   // only M is clamped
        else { tmin[0] = min(tmin[0], k); tmax[0] = max(tmax[0], k); }
        wsw[o_m + k] = mv;
      }
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const int mn = wave_min(tmin[c]), mxk = wave_max(tmax[c]);
        const bool has = (c == 0) || (c == 1 && has_i1) || (c == 2 && has_d1) || (NCOMP == 5 && c == 3 && has_i2) || (NCOMP == 5 && c == 4 && has_d2);
        if (has && mn != INT_MAX) { tlo[c] = mn; thi[c] = mxk; }
      }
      sd.cur_exists = 1; sd.cur_lo = tlo[0]; sd.cur_hi = thi[0]; sd.cur_idx0 = o_m;
    }
  }
  __syncthreads();   // every lane has read the inputs' directory records before the slot of score s is overwritten
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) { mslot[MT::LO + c] = tlo[c]; mslot[MT::HI + c] = thi[c]; }
    mslot[MT::BASE] = base; mslot[MT::WIDTH] = width; mslot[MT::DATA] = data; mslot[MT::EXISTS] = exists;
    if (sd.dir) { int* d = sd.dir + (long long)s * MT::INTS; for (int c = 0; c < MT::INTS; ++c) d[c] = mslot[c]; }
  }
  __syncthreads();
  return ok;
}

struct BiBreakpoint {
  int score, score_forward, score_reverse, k_forward, k_reverse, offset_forward, offset_reverse, component;
};

// R/wavefront_bialign.c:189-311: component c of aligner 0 at score_0 against the same component of aligner 1 at score_1;
// the lowest diagonal of aligner 0 on which the two offsets meet (and, for I/D, lie inside the matrix) wins.
template <int NCOMP>
__device__ __forceinline__ void bi_breakpoint_cc(const BiSide<NCOMP>& s0, const BiSide<NCOMP>& s1, const int* m0, const int* m1,
                                                 const WfaDevConfig& cfg, bool forward, int score_0, int score_1, int c,
                                                 int plen, int tlen, BiBreakpoint& bp, int lane) {
  typedef Meta<NCOMP> MT;
  const int gap_open = (c == 0) ? 0 : ((c == 1 || c == 2) ? cfg.o1 : cfg.o2);
  const int lo_0 = m0[MT::LO + c], hi_0 = m0[MT::HI + c];
  const int lo_1 = tlen - plen - m1[MT::HI + c], hi_1 = tlen - plen - m1[MT::LO + c];
  if (hi_1 < lo_0 || hi_0 < lo_1) return;
  if (score_0 + score_1 - gap_open >= bp.score) return;
  const int min_hi = min(hi_0, hi_1), max_lo = max(lo_0, lo_1);
  const int i0 = m0[MT::DATA] + c * m0[MT::WIDTH] - m0[MT::BASE], i1 = m1[MT::DATA] + c * m1[MT::WIDTH] - m1[MT::BASE];
  for (int kb = max_lo; kb <= min_hi; kb += 64) {
    const int k_0 = kb + lane;
    bool hit = false;
    int o0 = 0, o1 = 0;
    if (k_0 <= min_hi) {
      const int k_1 = tlen - plen - k_0;
      o0 = s0.ws[i0 + k_0]; o1 = s1.ws[i1 + k_1];
      hit = (long long)o0 + o1 >= tlen;
      if (hit && c != 0) {   // interior I/D offsets may lie outside the matrix (they are not clamped): skipped (:222-226,236-240)
        const int kk = forward ? k_0 : k_1, oo = forward ? o0 : o1;
        if (oo - kk > plen || oo > tlen) hit = false;
      }
    }
    const unsigned long long bm = __ballot(hit);
    if (bm) {
      const int L = __builtin_ctzll(bm);
      const int fk0 = kb + L, fk1 = tlen - plen - fk0;
      const int fo0 = __builtin_amdgcn_readlane(o0, L), fo1 = __builtin_amdgcn_readlane(o1, L);
      if (forward) {
        bp.score_forward = score_0; bp.score_reverse = score_1; bp.k_forward = fk0; bp.k_reverse = fk1;
        bp.offset_forward = fo0; bp.offset_reverse = fo1;
      } else {
        bp.score_forward = score_1; bp.score_reverse = score_0; bp.k_forward = fk1; bp.k_reverse = fk0;
        bp.offset_forward = fo1; bp.offset_reverse = fo0;
      }
      bp.score = score_0 + score_1 - gap_open;
      bp.component = c;
      return;
    }
  }
}

// R/wavefront_bialign.c:315-395 (wavefront_bialign_overlap)
template <int NCOMP>
__device__ __forceinline__ void bi_overlap(const BiSide<NCOMP>& s0, const BiSide<NCOMP>& s1, const WfaDevConfig& cfg, int scope,
                                           int score_0, int score_1, bool forward, int plen, int tlen, BiBreakpoint& bp, int lane) {
  typedef Meta<NCOMP> MT;
  const int* m0 = s0.ring + (score_0 % scope) * MT::INTS;
  if (!m0[MT::EXISTS]) return;
  for (int i = 0; i < scope; ++i) {
    const int score_i = score_1 - i;
    if (score_i < 0) break;
    const int* m1 = s1.ring + (score_i % scope) * MT::INTS;
    if (NCOMP == 5) {
      if (score_0 + score_i - cfg.o2 >= bp.score) continue;
      bi_breakpoint_cc<NCOMP>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 4, plen, tlen, bp, lane);
      bi_breakpoint_cc<NCOMP>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 3, plen, tlen, bp, lane);
    }
    if (NCOMP >= 3) {
      if (score_0 + score_i - cfg.o1 >= bp.score) continue;
      bi_breakpoint_cc<NCOMP>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 2, plen, tlen, bp, lane);
      bi_breakpoint_cc<NCOMP>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 1, plen, tlen, bp, lane);
    }
    if (score_0 + score_i >= bp.score) continue;
    if (m1[MT::EXISTS]) bi_breakpoint_cc<NCOMP>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 0, plen, tlen, bp, lane);
  }
}

// candidate of the base-case backtrace (R/wavefront_backtrace.c:64-219) from the HBM directory of the base aligner
template <int NCOMP>
__device__ __forceinline__ long long bi_bt_cand(const BiSide<NCOMP>& sd, int s, int c, int k, int add, int type) {
  typedef Meta<NCOMP> MT;
  if (s < 0) return WFA_OFFSET_NULL;
  const int* m = sd.dir + (long long)s * MT::INTS;
  if (k < m[MT::LO + c] || k > m[MT::HI + c]) return WFA_OFFSET_NULL;
  const int o = sd.ws[m[MT::DATA] + c * m[MT::WIDTH] + (k - m[MT::BASE])];
  return (((long long)(o + add)) << 4) | type;
}

// R/wavefront_backtrace.c:320-529 (and :223-319 for the single-component metrics) with begin / end components; one lane;
// ops written right to left, ending at buf + end_pos.  Returns the index of the first op.
template <int NCOMP>
__device__ long long bi_backtrace(const BiSide<NCOMP>& sd, const WfaDevConfig& cfg, int plen, int tlen, int end_s, int end_k,
                                  int end_off, int comp_end, uint8_t* buf, long long end_pos) {
  enum { BT_I1_OPEN = 1, BT_I1_EXT, BT_I2_OPEN, BT_I2_EXT, BT_D1_OPEN, BT_D1_EXT, BT_D2_OPEN, BT_D2_EXT, BT_M };
  long long begin = end_pos;
  auto push = [&](char c, int n) { while (n-- > 0) buf[--begin] = (uint8_t)c; };
  int comp = comp_end, s = end_s, k = end_k, offset = end_off;
  int h = offset, v = offset - k;
  if (comp_end == 0) {
    if (v < plen) push('D', plen - v);
    if (h < tlen) push('I', tlen - h);
  }
  const int oe1 = (NCOMP == 1) ? cfg.o1 : cfg.o1 + cfg.e1;
  while (v > 0 && h > 0 && s > 0) {
    const int s_x = s - cfg.x, s_o1 = s - oe1, s_e1 = s - cfg.e1;
    const int s_o2 = s - cfg.o2 - cfg.e2, s_e2 = s - cfg.e2;
    long long best;
    if (comp == 0) {
      best = (NCOMP == 1 && cfg.metric == 0) ? (long long)WFA_OFFSET_NULL : bi_bt_cand<NCOMP>(sd, s_x, 0, k, 1, BT_M);
      best = max(best, bi_bt_cand<NCOMP>(sd, s_o1, 0, k - 1, 1, BT_I1_OPEN));
      best = max(best, bi_bt_cand<NCOMP>(sd, s_o1, 0, k + 1, 0, BT_D1_OPEN));
      if (NCOMP >= 3) {
        best = max(best, bi_bt_cand<NCOMP>(sd, s_e1, 1, k - 1, 1, BT_I1_EXT));
        best = max(best, bi_bt_cand<NCOMP>(sd, s_e1, 2, k + 1, 0, BT_D1_EXT));
      }
      if (NCOMP == 5) {
        best = max(best, bi_bt_cand<NCOMP>(sd, s_o2, 0, k - 1, 1, BT_I2_OPEN));
        best = max(best, bi_bt_cand<NCOMP>(sd, s_e2, 3, k - 1, 1, BT_I2_EXT));
        best = max(best, bi_bt_cand<NCOMP>(sd, s_o2, 0, k + 1, 0, BT_D2_OPEN));
        best = max(best, bi_bt_cand<NCOMP>(sd, s_e2, 4, k + 1, 0, BT_D2_EXT));
      }
    } else if (comp == 1) {
      best = max(bi_bt_cand<NCOMP>(sd, s_o1, 0, k - 1, 1, BT_I1_OPEN), bi_bt_cand<NCOMP>(sd, s_e1, NCOMP >= 3 ? 1 : 0, k - 1, 1, BT_I1_EXT));
    } else if (comp == 2) {
      best = max(bi_bt_cand<NCOMP>(sd, s_o1, 0, k + 1, 0, BT_D1_OPEN), bi_bt_cand<NCOMP>(sd, s_e1, NCOMP >= 3 ? 2 : 0, k + 1, 0, BT_D1_EXT));
    } else if (comp == 3) {
      best = max(bi_bt_cand<NCOMP>(sd, s_o2, 0, k - 1, 1, BT_I2_OPEN), bi_bt_cand<NCOMP>(sd, s_e2, NCOMP == 5 ? 3 : 0, k - 1, 1, BT_I2_EXT));
    } else {
      best = max(bi_bt_cand<NCOMP>(sd, s_o2, 0, k + 1, 0, BT_D2_OPEN), bi_bt_cand<NCOMP>(sd, s_e2, NCOMP == 5 ? 4 : 0, k + 1, 0, BT_D2_EXT));
    }
    if (best < 0) break;
    if (comp == 0) {
      const int src = (int)(best >> 4);
      push('M', offset - src);
      offset = src;
      v = offset - k; h = offset;
      if (v <= 0 || h <= 0) break;
    }
    const int type = (int)(best & 0xF);
    switch (type) {
      case BT_M: s = s_x; comp = 0; break;
      case BT_I1_OPEN: s = s_o1; comp = 0; break;
      case BT_I1_EXT: s = s_e1; comp = 1; break;
      case BT_I2_OPEN: s = s_o2; comp = 0; break;
      case BT_I2_EXT: s = s_e2; comp = 3; break;
      case BT_D1_OPEN: s = s_o1; comp = 0; break;
      case BT_D1_EXT: s = s_e1; comp = 2; break;
      case BT_D2_OPEN: s = s_o2; comp = 0; break;
      default: s = s_e2; comp = 4; break;
    }
    if (type == BT_M) { push('X', 1); --offset; }
    else if (type <= BT_I2_EXT) { push('I', 1); --k; --offset; }
    else { push('D', 1); ++k; }
    v = offset - k; h = offset;
  }
  if (comp == 0) {
    if (v > 0 && h > 0) { const int n = min(v, h); push('M', n); v -= n; h -= n; }
    push('D', max(v, 0));
    push('I', max(h, 0));
  }
  return begin;
}

template <int NCOMP, bool PACKED>
__global__ void __launch_bounds__(64)
wfa_biwfa_kernel(const BiwfaArgs a) {
  typedef Meta<NCOMP> MT;
  extern __shared__ int smem[];
  const WfaDevConfig& cfg = a.k.cfg;
  const int scope = cfg.scope;
  const int lane = threadIdx.x;
  int* const ring_f = smem;
  int* const ring_r = ring_f + scope * MT::INTS;
  int* const ring_b = ring_r + scope * MT::INTS;
  int* const stack = ring_b + scope * MT::INTS;       // WFA_BI_STACK windows of 8 ints
  int* const wsb = a.k.ws + (long long)blockIdx.x * a.k.ws_stride;
  const uint32_t nwork = a.k.nwork_dev ? *a.k.nwork_dev : a.k.nwork;
  const long long max_steps = cfg.max_steps;

  for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t pair = a.k.worklist ? a.k.worklist[wi] : wi;
    const WfaPairMeta pm = a.k.meta[pair];
    BiView<PACKED> view;
    view.wildcard = cfg.wildcard;
    if (PACKED) { view.pw = a.k.words + pm.p_woff; view.tw = a.k.words + pm.t_woff; view.pb = nullptr; view.tb = nullptr; }
    else { view.pb = a.k.bytes + a.k.p_boff[pair]; view.tb = a.k.bytes + a.k.t_boff[pair]; view.pw = nullptr; view.tw = nullptr; }
    uint8_t* const out = a.score_only ? nullptr : a.k.cigar_ops + a.k.cigar_off[pair];
    long long out_len = 0;
    int status = 0, top_score = INT_MIN;   // INT_MIN: no top-level split (the score stays unset, Q6)
    bool hand_on = false;                  // the top-level base case outgrew its history: the general kernel takes the pair
    BiSide<NCOMP> F, R, B;
    F.ring = ring_f; F.ws = wsb; F.dir = nullptr; F.stride = a.ring_stride; F.slots = scope;
    R.ring = ring_r; R.ws = wsb + a.ring_ints; R.dir = nullptr; R.stride = a.ring_stride; R.slots = scope;
    B.ring = ring_b; B.ws = wsb + 2 * a.ring_ints; B.stride = a.base_stride; B.slots = WFA_BI_BASE_SLOTS;
    B.dir = wsb + 2 * a.ring_ints + a.base_ints - (long long)WFA_BI_BASE_SLOTS * MT::INTS;
    // the recursion, depth first: a window is {pbeg, pend, tbeg, tend, begin comp | end comp << 4 | level-0 flag << 8 |
    // ends-free form << 9, score_remaining}
    int sp = 0;
    __syncthreads();
    if (lane == 0) {
      int* w = stack;
      w[0] = 0; w[1] = pm.plen; w[2] = 0; w[3] = pm.tlen;
      w[4] = 0 | (0 << 4) | (1 << 8) | ((cfg.endsfree ? 1 : 0) << 9);
      w[5] = (!a.score_only && max(pm.plen, pm.tlen) <= WFA_BI_FALLBACK_MIN_LENGTH) ? 0 : INT_MAX;
    }
    sp = 1;
    __syncthreads();
    while (sp > 0 && status == 0) {
      --sp;
      const int* w = stack + sp * 8;
      const int pbeg = w[0], pend = w[1], tbeg = w[2], tend = w[3], flags = w[4], score_remaining = w[5];
      const int comp_begin = flags & 15, comp_end = (flags >> 4) & 15;
      const bool level0 = (flags >> 8) & 1, ef_form = (flags >> 9) & 1;
      const int plen = pend - pbeg, tlen = tend - tbeg;
      __syncthreads();   // the window has been read: its stack slot may be reused
      if (!a.score_only) {   // (the score-only form runs the breakpoint search whatever the lengths, :662-702)
        if (tlen == 0) { for (int i = lane; i < plen; i += 64) out[out_len + i] = 'D'; out_len += plen; continue; }
        if (plen == 0) { for (int i = lane; i < tlen; i += 64) out[out_len + i] = 'I'; out_len += tlen; continue; }
      }
      view.pbeg = pbeg; view.pend = pend; view.tbeg = tbeg; view.tend = tend;
      bool do_base = score_remaining <= WFA_BI_FALLBACK_MIN_SCORE;
      BiBreakpoint bp;
      bp.score = INT_MAX; bp.score_forward = 0; bp.score_reverse = 0; bp.k_forward = 0; bp.k_reverse = 0;
      bp.offset_forward = 0; bp.offset_reverse = 0; bp.component = 0;
      if (!do_base) {
        // ---------------- R/wavefront_bialign.c:411-519 (wavefront_bialign_find_breakpoint) ----------------
        int st = WFA_BI_OK, reached = 0;
        view.reverse = false;
        BiView<PACKED> rview = view; rview.reverse = true;
        bi_side_init<NCOMP>(F, scope, comp_begin, plen, tlen, lane);
        bi_side_init<NCOMP>(R, scope, comp_end, plen, tlen, lane);
        F.steps_wait = R.steps_wait = cfg.steps_between;   // (R/wavefront_heuristic.c:114-121)
        const int max_antidiagonal = plen + tlen - 1;
        int score_f = 0, score_r = 0;
        // one turn of an aligner: extend, end test; returns true when that aligner is done
        auto turn = [&](BiSide<NCOMP>& sd, const BiView<PACKED>& vw, int s, int cend, int* max_ak) -> bool {
          if (!sd.cur_exists) {
            *max_ak = 0;
            if (sd.null_steps > scope) { st = WFA_BI_END_UNREACHABLE; reached = s; return true; }
            return false;
          }
          const int best = bi_side_extend<NCOMP, PACKED>(sd, vw, plen, tlen, lane);
          if (bi_side_terminated<NCOMP>(sd, scope, s, cend, plen, tlen)) { st = WFA_BI_END_REACHED; reached = s; *max_ak = 0; return true; }
          bi_side_cutoff<NCOMP>(sd, cfg, scope, s, plen, tlen, lane);
          *max_ak = best;
          return false;
        };
        int f_max_ak = 0, r_max_ak = 0, max_ak = 0;
        bool quit = turn(F, view, 0, comp_end, &f_max_ak);
        if (!quit) quit = turn(R, rview, 0, comp_begin, &r_max_ak);
        bool last_forward = false;
        while (!quit) {
          if (f_max_ak + r_max_ak >= max_antidiagonal) break;
          ++score_f;
          bi_side_compute<NCOMP>(F, cfg, scope, score_f, plen, tlen, lane);
          quit = turn(F, view, score_f, comp_end, &max_ak);
          if (f_max_ak < max_ak) f_max_ak = max_ak;
          last_forward = true;
          if (quit) break;
          if (f_max_ak + r_max_ak >= max_antidiagonal) break;
          ++score_r;
          bi_side_compute<NCOMP>(R, cfg, scope, score_r, plen, tlen, lane);
          quit = turn(R, rview, score_r, comp_begin, &max_ak);
          if (r_max_ak < max_ak) r_max_ak = max_ak;
          last_forward = false;
          if (quit) break;
          if ((long long)score_r + score_f >= max_steps) { st = WFA_STATUS_MAX_STEPS_REACHED; quit = true; }
        }
        if (!quit) {
          const int gap_opening = (NCOMP == 3) ? cfg.o1 : (NCOMP == 5) ? max(cfg.o1, cfg.o2) : 0;
          while (true) {
            if (last_forward) {
              const int min_score_reverse = (score_r > scope - 1) ? score_r - (scope - 1) : 0;
              if (score_f + min_score_reverse - gap_opening >= bp.score) break;
              bi_overlap<NCOMP>(F, R, cfg, scope, score_f, score_r, true, plen, tlen, bp, lane);
              ++score_r;
              bi_side_compute<NCOMP>(R, cfg, scope, score_r, plen, tlen, lane);
              if (turn(R, rview, score_r, comp_begin, &max_ak)) { quit = true; break; }
            }
            const int min_score_forward = (score_f > scope - 1) ? score_f - (scope - 1) : 0;
            if (min_score_forward + score_r - gap_opening >= bp.score) break;
            bi_overlap<NCOMP>(R, F, cfg, scope, score_r, score_f, false, plen, tlen, bp, lane);
            ++score_f;
            bi_side_compute<NCOMP>(F, cfg, scope, score_f, plen, tlen, lane);
            if (turn(F, view, score_f, comp_end, &max_ak)) { quit = true; break; }
            if ((long long)score_r + score_f >= max_steps) { st = WFA_STATUS_MAX_STEPS_REACHED; quit = true; break; }
            last_forward = true;
          }
        }
        if (a.score_only) {
          // R/wavefront_bialign.c:683-701: a breakpoint, or an end reached before any overlap, is a completed alignment with
          // that score; anything else leaves the score unset
          if (!quit) top_score = bp.score;
          else if (st == WFA_BI_END_REACHED) top_score = reached;
          else status = (st == WFA_STATUS_MAX_STEPS_REACHED) ? WFA_STATUS_MAX_STEPS_REACHED : WFA_STATUS_UNATTAINABLE;
          break;
        }
        if (quit) {
          // R/wavefront_bialign.c:520-548 (wavefront_bialign_find_breakpoint_exception)
          if (st == WFA_BI_END_REACHED && reached <= WFA_BI_RECOVERY_MIN_SCORE) do_base = true;
          else { status = (st == WFA_STATUS_MAX_STEPS_REACHED) ? WFA_STATUS_MAX_STEPS_REACHED : WFA_STATUS_UNATTAINABLE; break; }
        }
      }
      if (do_base) {
        // ---------------- R/wavefront_bialign.c:155-188 (wavefront_bialign_base): the ordinary algorithm ----------------
        view.reverse = false;
        B.stride = min(a.base_stride, plen + tlen + 3);
        bi_side_init<NCOMP>(B, scope, comp_begin, plen, tlen, lane);
        int s = 0, end_k = 0, end_off = 0;
        bool reached_end = false, fail = false;
        while (true) {
          if (!B.cur_exists) {
            if (B.null_steps > scope) { fail = true; break; }
          } else if (ef_form) {
            // R/wavefront_extend.c:263-297 + R/wavefront_termination.c:115-162 with all free ends 0: the lowest diagonal that
            // reaches the end of both sequences — only tlen - plen can
            bi_side_extend<NCOMP, PACKED>(B, view, plen, tlen, lane);
            const int ak = tlen - plen;
            if (B.cur_lo <= ak && ak <= B.cur_hi && B.ws[B.cur_idx0 + ak] >= tlen) { reached_end = true; end_k = ak; end_off = tlen; break; }
          } else {
            bi_side_extend<NCOMP, PACKED>(B, view, plen, tlen, lane);
            if (bi_side_terminated<NCOMP>(B, scope, s, comp_end, plen, tlen)) { reached_end = true; end_k = tlen - plen; end_off = tlen; break; }
          }
          ++s;
          if (s >= max_steps) { fail = true; break; }   // the base aligner's own limit: not "completed" (R/wavefront_bialign.c:182-187)
          // A window handed to the base case scores <= 250 (or <= 500 after an early end): only the top-level base case of
          // reads <= 100 bases has no bound (divergent pairs under large penalties).  Such a pair goes to the general kernel.
          if (s >= WFA_BI_BASE_SLOTS - 1) { hand_on = true; break; }
          if (!bi_side_compute<NCOMP>(B, cfg, scope, s, plen, tlen, lane)) { hand_on = true; break; }
        }
        if (hand_on) break;
        if (fail || !reached_end) { status = WFA_STATUS_UNATTAINABLE; break; }
        // backtrace right to left into the free tail of the pair's region, then move it down to out_len
        __syncthreads();
        long long begin = 0;
        const long long end_pos = out_len + plen + tlen;
        if (lane == 0) begin = bi_backtrace<NCOMP>(B, cfg, plen, tlen, s, end_k, end_off, comp_end, out, end_pos);
        begin = ((long long)__builtin_amdgcn_readfirstlane((int)(begin >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)begin);
        __syncthreads();
        const long long n = end_pos - begin;
        if (begin != out_len) {
          for (long long i0 = 0; i0 < n; i0 += 64) {
            uint8_t c = 0;
            if (i0 + lane < n) c = out[begin + i0 + lane];
            __syncthreads();
            if (i0 + lane < n) out[out_len + i0 + lane] = c;
            __syncthreads();
          }
        }
        out_len += n;
        continue;
      }
      // ---------------- breakpoint found: the two halves (R/wavefront_bialign.c:614-650) ----------------
      const int bh = bp.offset_forward, bv = bp.offset_forward - bp.k_forward;
      if (level0) top_score = bp.score;
      if (sp + 2 > WFA_BI_STACK) { status = WFA_STATUS_OOM; break; }
      if (lane == 0) {
        int* w1 = stack + sp * 8;         // second half: processed after the first
        w1[0] = pbeg + bv; w1[1] = pend; w1[2] = tbeg + bh; w1[3] = tend;
        w1[4] = bp.component | (comp_end << 4); w1[5] = bp.score_reverse;
        int* w0 = stack + (sp + 1) * 8;   // first half: on top
        w0[0] = pbeg; w0[1] = pbeg + bv; w0[2] = tbeg; w0[3] = tbeg + bh;
        w0[4] = comp_begin | (bp.component << 4); w0[5] = bp.score_forward;
      }
      sp += 2;
      __syncthreads();
    }
    if (lane == 0) {
      int out_score = INT_MIN, out_status = status;
      if (hand_on && a.k.fb_list) {
        out_status = WFA_INTERNAL_FALLBACK; out_len = 0;
        a.k.fb_list[atomicAdd(a.k.fb_count, 1u)] = pair;
      } else if (hand_on) {
        out_status = WFA_STATUS_UNATTAINABLE; out_len = 0;
      } else if (status == 0) {
        if (top_score != INT_MIN) out_score = classic_score(cfg, pm.plen, pm.tlen, top_score);
      } else if (status == WFA_STATUS_OOM) {
        out_len = 0;
      }   // (step limit / unattainable: the ops appended so far stay, as in the reference's cigar)
      a.k.score[pair] = out_score;
      a.k.status[pair] = out_status;
      if (!a.score_only) {
        a.k.cigar_begin[pair] = a.k.cigar_off[pair];
        a.k.cigar_len[pair] = (int)out_len;
      }
    }
    __syncthreads();
  }
}

// host entry points (csrc/k_biwfa.hip, one translation unit per component count)
int launch_biwfa_c1(bool packed, const BiwfaArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_biwfa_c3(bool packed, const BiwfaArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_biwfa_c5(bool packed, const BiwfaArgs& a, int grid, size_t smem, hipStream_t stream);

template <int NCOMP>
inline int launch_biwfa_ncomp(bool packed, const BiwfaArgs& a, int grid, size_t smem, hipStream_t stream) {
  if (packed) hipLaunchKernelGGL((wfa_biwfa_kernel<NCOMP, true>), dim3(grid), dim3(64), smem, stream, a);
  else hipLaunchKernelGGL((wfa_biwfa_kernel<NCOMP, false>), dim3(grid), dim3(64), smem, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// LDS bytes of a workgroup: three directory rings + the window stack
inline size_t biwfa_smem(int ncomp, int scope) {
  return ((size_t)3 * scope * (2 * ncomp + 4) + (size_t)WFA_BI_STACK * 8 + 8) * sizeof(int);
}
// diagonals a base-case wavefront can span (scores <= 500 either way of diagonal 0) — or the whole matrix of short reads
inline int biwfa_base_stride(int max_width) { return std::min(2 * (WFA_BI_RECOVERY_MIN_SCORE + 1) + 3, max_width); }
inline int64_t biwfa_base_ints(int ncomp, int base_stride) {
  return (int64_t)WFA_BI_BASE_SLOTS * ncomp * base_stride + (int64_t)WFA_BI_BASE_SLOTS * (2 * ncomp + 4) + 64;
}

inline int launch_biwfa_any(int ncomp, bool packed, const BiwfaArgs& a, int grid, hipStream_t stream) {
  const size_t smem = biwfa_smem(ncomp, a.k.cfg.scope);
  if (ncomp == 1) return launch_biwfa_c1(packed, a, grid, smem, stream);
  if (ncomp == 3) return launch_biwfa_c3(packed, a, grid, smem, stream);
  return launch_biwfa_c5(packed, a, grid, smem, stream);
}

}  // namespace wfa
