// wfa_bilevel.hpp — BiWFA, level by level (round 5): the recursion of R/wavefront_bialign.c:581-658 run BREADTH first, so that
// every window of a recursion level is a work item of one launch and the chip is full at every depth — wfa_biwfa.hpp's kernel
// (one wave per alignment, depth first) spends five of six levels on windows a few chunks wide with one wave per SIMD or less.
//
//   bl_seed_kernel    one window per pair (the whole pair), results reset
//   bl_split_kernel   one workgroup per window of level l: the breakpoint search (forward + reverse score-only rings, overlap,
//                     R/wavefront_bialign.c:411-519) -> two child windows.  A child whose score is <= 250 goes to the base queue,
//                     an empty / one-sided child is written out at once, the others are level l + 1.  Wide levels run 256 threads
//                     per window (chunks of the wavefront on four waves), deep levels 64.
//   bl_base_kernel    one wave per base window: the ordinary algorithm with a history and a backtrace (R/wavefront_bialign.c:155-188)
//   bl_finish_kernel  one wave per pair: the op strings of its leaves, in sequence order, moved together; score and status
//
// Why breadth first gives the same result.  A window's search reads nothing but its own sub-sequences, begin / end component and
// score bound (the heuristic state is re-set per search, R/wavefront_heuristic.c:114-121), so the set of windows and each window's
// op string do not depend on the visiting order.  Windows tile the pair: the ops of window [pbeg,pend) x [tbeg,tend) are written
// into bytes [pbeg + tbeg, pend + tend) of the pair's region (an op consumes at least one base), and bl_finish_kernel concatenates
// the leaves by their start.  Failures (step limit, unattainable): depth first stops at the first failing window in sequence order
// with everything left of it complete; here the smallest start among failing windows is kept per pair and the leaves right of it
// are dropped (R/wavefront_bialign.c:614-650: the ops appended so far stay).
// What does not fit (a queue or the leaf list of a pair is full, the leaves do not tile the pair) is aligned again by
// wfa_biwfa_kernel (redo list): capacity is a speed matter, never a result.
//
// Offsets of the rings: int16 when the sequences allow (stored clamped to [-16384, 32767]: a negative offset is dead, one beyond
// the text stays beyond it — R/wavefront_offset.h:44-57), else int32.  compute-next and extend are one pass (the extended M is what
// is stored, as R/wavefront_extend.c leaves it), two chunks of loads in flight per thread.
#pragma once
#include "wfa_biwfa.hpp"

namespace wfa {

struct BlWindow { int pair, pbeg, pend, tbeg, tend, flags, score_remaining, pad; };
struct BlLeaf { int start, region, begin, n, next, pad0, pad1, pad2; };   // begin: relative to the pair's op region

#ifndef WFA_BL_FLY
#define WFA_BL_FLY 2      // chunks of loads a thread of the workspace form keeps in flight (4: 128 registers, four waves per SIMD — measured slower)
#endif
#define WFA_BL_MAX_LEVELS 40
#define WFA_BL_LEAF_LDS 1024          // leaves of a pair bl_finish_kernel sorts in LDS (more: redo)
#define WFA_BL_FLAG_REDO 1
#define WFA_BL_FLAG_HANDON 2

struct BlArgs {
  WfaKernelArgs k;
  BlWindow* q[2];          // windows of the even / odd levels
  BlWindow* qb;            // base windows
  BlWindow* qw;            // windows of the current level that outgrew the LDS form (taken by the workspace form of the same level)
  BlLeaf* leaves;
  uint32_t* cnt;           // [l] windows of level l (l < WFA_BL_MAX_LEVELS), [40] base windows, [41] leaves, [42] redo pairs; [64 + l], [64 + 40]: windows taken; [128 + l]: windows of level l handed to the workspace form, [192 + l]: taken
#define WFA_BL_COUNTER_WORDS 256
  uint32_t qcap, qbcap, leafcap, qwcap;
  int* head;               // per pair: first leaf of its list (-1: none)
  int* flags;              // per pair: WFA_BL_FLAG_*
  int* top;                // per pair: score of the top-level breakpoint (INT_MIN: none)
  unsigned long long* failkey;   // per pair: (start << 32 | status) of the first failing window in sequence order
  uint32_t* redo_list;
  void* rings;             // slice of workgroup b: rings + b * slice_bytes (forward ring, reverse ring) / base history
  long long slice_bytes;
  long long ring_elems;    // elements of ONE ring
  int ring_stride;         // diagonals per component row of a ring at most (plen + tlen + 3 of the longest pair)
  long long base_ints;
  int base_stride;
  int level;
  int from_wide;           // 1: the launch reads qw (count cnt[128 + level]) instead of q[level & 1]
  int lds_w, lds_slots, lds_seq_words;   // LDS form: diagonals per row (a power of two), rows per component, words per sequence buffer
};

// pointer to ring offsets / packed words: generic (workspace) or LDS (address space 3: ds_read / ds_write instead of flat accesses)
template <typename T, bool LDSR> struct BlPtr { typedef T* type; };
template <typename T> struct BlPtr<T, true> { typedef __attribute__((address_space(3))) T* type; };

template <typename P> __device__ __forceinline__ int bl_ld(P p, int i) {
  int v = p[i];
  if (sizeof(p[0]) == 2) v = (v < 0) ? WFA_OFFSET_NULL : v;
  return v;
}
// the step's own loads: int16 rows hold -16384 for every dead cell (bl_st), which max / + 1 / the bound tests treat as NULL
template <typename P> __device__ __forceinline__ int bl_ld_raw(P p, int i) { return p[i]; }
template <typename P> __device__ __forceinline__ void bl_st(P p, int i, int v) {
  if (sizeof(p[0]) == 2) p[i] = (short)((v < 0) ? -16384 : min(v, 32767));
  else p[i] = v;
}

// N values reduced over the workgroup: bit i of maxmask set = maximum, else minimum.  One wave: shuffles.  More: the waves' partial
// results meet in LDS atomics — red holds three buffers of 16 minima + 16 maxima used in rotation (`phase`), the buffer of the call
// after the next is re-set before this call's barrier, so a call costs ONE barrier whatever the number of waves.
template <int THREADS, int N>
__device__ __forceinline__ void bl_reduce(int (&v)[N], uint32_t maxmask, int* red, int& phase, int tid, uint32_t wave_uniform = 0u) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    // (wave_uniform: lane 0 — the lane that stays longest in the chunk loop — holds the wave's value already)
    if ((wave_uniform >> i) & 1) v[i] = __builtin_amdgcn_readfirstlane(v[i]);
    else v[i] = ((maxmask >> i) & 1) ? wave_max(v[i]) : wave_min(v[i]);
  }
  if (THREADS > 64) {
    int* const cur = red + phase * 32;
    int* const nxt = red + ((phase + 1) % 3) * 32;
    if (tid < 32) nxt[tid] = (tid < 16) ? INT_MAX : INT_MIN;
    if ((tid & 63) == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) { if ((maxmask >> i) & 1) atomicMax(cur + 16 + i, v[i]); else atomicMin(cur + i, v[i]); }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ((maxmask >> i) & 1) ? cur[16 + i] : cur[i];
    phase = (phase + 1) % 3;
  }
}
__device__ __forceinline__ void bl_reduce_init(int* red, int tid) { if (tid < 96) red[tid] = ((tid & 31) < 16) ? INT_MAX : INT_MIN; }

// a window of the two sequences, read forwards or backwards (R/wavefront_sequences.c:275-310): wfa_biwfa.hpp's BiView over either
// pointer kind
template <bool PACKED, bool LDSR>
struct BlView {
  typedef typename BlPtr<const uint32_t, LDSR>::type WP;
  WP pw; WP tw;
  const uint8_t* pb; const uint8_t* tb;
  int pbeg, pend, tbeg, tend, wildcard;
  bool reverse;
  static __device__ __forceinline__ uint32_t w16(WP w, int pos) {
    const int i = pos >> 4;
    const uint32_t lo = w[i], hi = w[i + 1];
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(pos & 15) << 1);
  }
  __device__ __forceinline__ int run(int v, int h, int maxrun) const {
    int n = 0;
    if (PACKED) {
      if (!reverse) {
        const int pv = pbeg + v, th = tbeg + h;
        while (n < maxrun) {
          const uint32_t x = w16(pw, pv + n) ^ w16(tw, th + n);
          const int m = x ? (__builtin_ctz(x) >> 1) : 16;
          n += m;
          if (m < 16) break;
        }
      } else {
        const int pq = pend - 1 - v, tq = tend - 1 - h;
        while (n < maxrun) {
          const int a = pq - n, b = tq - n;
          const int sa = (a >= 15) ? 0 : 15 - a, sb = (b >= 15) ? 0 : 15 - b;
          const uint32_t wa = w16(pw, a - 15 + sa) << (2 * sa), wb = w16(tw, b - 15 + sb) << (2 * sb);
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

// One unidirectional aligner of the workgroup: directory ring in LDS; the offsets in the workgroup's slice of the workspace (rows of
// `stride` diagonals, element k - rbase, row of score s = s mod scope) or, LDSR, in LDS (rows of W diagonals, element k mod W, the row
// of a score taken from a counter of the non-null steps: scope / g + 1 rows per component are enough, scores between multiples of g
// being null).  Either way element (row, k) = idx0 + (k & kmask) with the row's idx0 kept in the directory record.
template <int NCOMP, typename OT, bool LDSR>
struct BlSide {
  typedef typename BlPtr<OT, LDSR>::type P;
  int* ring;
  P ws;
  int stride, rbase, kmask;
  int lds_slots, cur_slot;
  int g;             // bl_lattice_gcd
  int null_steps;
  int cur_lo, cur_hi, cur_exists, end_reached;
  int cur_idx0;
  int steps_wait, have_max_sw, max_sw;
  __device__ __forceinline__ int next_data(int s, int scope) {
    if (LDSR) { cur_slot = (cur_slot + 1 == lds_slots) ? 0 : cur_slot + 1; return cur_slot * NCOMP * stride; }
    return (s % scope) * NCOMP * stride;
  }
};

template <int NCOMP, typename OT, bool LDSR>
__device__ __forceinline__ void bl_side_init(BlSide<NCOMP, OT, LDSR>& sd, int scope, int comp_begin, int plen, int tlen, int tid) {
  typedef Meta<NCOMP> MT;
  sd.rbase = LDSR ? 0 : -plen - 1;
  sd.kmask = LDSR ? sd.stride - 1 : -1;
  sd.cur_slot = 0;
  sd.null_steps = 0; sd.end_reached = 0;
  sd.steps_wait = 0; sd.have_max_sw = 0; sd.max_sw = 0;
  const int data = 0;
  __syncthreads();
  if (tid == 0) {
    int* m = sd.ring;
    for (int c = 0; c < NCOMP; ++c) { m[MT::LO + c] = 1; m[MT::HI + c] = -1; }
    m[MT::LO + comp_begin] = 0; m[MT::HI + comp_begin] = 0;
    m[MT::BASE] = sd.rbase; m[MT::WIDTH] = sd.stride; m[MT::DATA] = data; m[MT::EXISTS] = (comp_begin == 0) ? 1 : 0;
    bl_st(sd.ws, data + comp_begin * sd.stride - sd.rbase, 0);
  }
  sd.cur_exists = (comp_begin == 0) ? 1 : 0;
  sd.cur_lo = sd.cur_exists ? 0 : 1; sd.cur_hi = sd.cur_exists ? 0 : -1;
  sd.cur_idx0 = data - sd.rbase;
  __syncthreads();
}

// extension of wavefront 0 (the later ones are extended inside bl_side_step)
template <int NCOMP, typename OT, int THREADS, bool LDSR, typename V>
__device__ __forceinline__ int bl_side_extend0(BlSide<NCOMP, OT, LDSR>& sd, const V& view, int plen, int tlen, int* red, int& phase, int tid) {
  int best[1] = {0};
  if (sd.cur_exists) {
    if (tid == 0) {
      const int off = bl_ld(sd.ws, sd.cur_idx0 + 0);
      const int ext = off + view.run(off, off, min(plen - off, tlen - off));
      if (ext != off) bl_st(sd.ws, sd.cur_idx0 + 0, ext);
      best[0] = 2 * ext;
    }
    bl_reduce<THREADS, 1>(best, 1u, red, phase, tid);
  }
  __syncthreads();
  return best[0];
}

template <int NCOMP, typename OT, bool LDSR>
struct BlIn {
  typedef typename BlPtr<OT, LDSR>::type P;
  int lo, hi, kmask, idx0;
  int klo; uint32_t span;   // the range as one unsigned test: k - klo <= span (a null wavefront: klo = 2^30, span = 0 — no diagonal passes)
  __device__ __forceinline__ bool null() const { return lo > hi; }
  __device__ __forceinline__ void set_null() { lo = 1; hi = -1; idx0 = 0; klo = 1 << 30; span = 0; }
  __device__ __forceinline__ int get(P ws, int k) const {
    return ((uint32_t)(k - klo) <= span) ? bl_ld_raw(ws, idx0 + (k & kmask)) : (sizeof(ws[0]) == 2 ? -16384 : WFA_OFFSET_NULL);
  }
};
template <int NCOMP, typename OT, bool LDSR>
__device__ __forceinline__ BlIn<NCOMP, OT, LDSR> bl_fetch_in(const BlSide<NCOMP, OT, LDSR>& sd, int scope, int s, int c) {
  typedef Meta<NCOMP> MT;
  BlIn<NCOMP, OT, LDSR> in;
  in.set_null(); in.kmask = sd.kmask;
  if (s >= 0) {
    const int* m = sd.ring + (s % scope) * MT::INTS;
    const int lo = m[MT::LO + c], hi = m[MT::HI + c];
    if (lo <= hi) { in.lo = lo; in.hi = hi; in.klo = lo; in.span = (uint32_t)(hi - lo); in.idx0 = m[MT::DATA] + c * m[MT::WIDTH] - m[MT::BASE]; }
  }
  return in;
}

// compute-next of score s and the extension of its M wavefront in one pass (R/wavefront_compute_affine.c:44-86,229-260,
// R/wavefront_compute_affine2p.c:45-106,334-368, R/wavefront_compute_edit.c / _linear.c, limits R/wavefront_compute.c:40-86,
// trimming :571-605 on the offsets before the extension, R/wavefront_extend.c:90-125), and the end test of R/wavefront_termination.c:
// 37-113 on the values as they pass (sd.end_reached).  Returns the largest antidiagonal 2 * offset - k of the extended M wavefront
// (0: none); -1: the wavefront does not fit the LDS rows (LDSR).
// scores are sums of the penalties the recurrences of this metric use: every other score is a null step (R/wavefront_compute.c:
// the reference walks them one by one; nothing is computed and nothing is read from them later)
template <int NCOMP>
__device__ __forceinline__ int bl_lattice_gcd(const WfaDevConfig& cfg) {
  auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
  int g = cfg.o1 + cfg.e1;
  if (!(NCOMP == 1 && cfg.metric == 0)) g = gcd(g, cfg.x);
  if (NCOMP != 1) g = gcd(g, cfg.e1);
  if (NCOMP == 5) { g = gcd(g, cfg.o2 + cfg.e2); g = gcd(g, cfg.e2); }
  return g > 0 ? g : 1;
}

template <int NCOMP, typename OT, int THREADS, bool LDSR, typename V>
__device__ __forceinline__ int bl_side_step(BlSide<NCOMP, OT, LDSR>& sd, const V& view, const WfaDevConfig& cfg, int scope, int s,
                                            int comp_end, int plen, int tlen, int* red, int& phase, int tid) {
  typedef Meta<NCOMP> MT;
  typedef BlIn<NCOMP, OT, LDSR> In;
  typedef typename BlPtr<OT, LDSR>::type P;
  if (sd.g > 1 && s % sd.g != 0) {   // off the lattice: a null step, registers only (its directory record is never read: bl_overlap skips these scores)
    ++sd.null_steps;
    sd.cur_exists = 0; sd.cur_lo = 1; sd.cur_hi = -1; sd.cur_idx0 = 0; sd.end_reached = 0;
    return 0;
  }
  const P ws = sd.ws;
  In nullin; nullin.set_null(); nullin.kmask = -1;
  const In mx = (NCOMP == 1 && cfg.metric == 0) ? nullin : bl_fetch_in<NCOMP, OT, LDSR>(sd, scope, s - cfg.x, 0);
  const In mo1 = bl_fetch_in<NCOMP, OT, LDSR>(sd, scope, s - cfg.o1 - cfg.e1, 0);
  const In i1e = (NCOMP == 1) ? nullin : bl_fetch_in<NCOMP, OT, LDSR>(sd, scope, s - cfg.e1, 1);
  const In d1e = (NCOMP == 1) ? nullin : bl_fetch_in<NCOMP, OT, LDSR>(sd, scope, s - cfg.e1, 2);
  In mo2 = nullin, i2e = nullin, d2e = nullin;
  if (NCOMP == 5) {
    mo2 = bl_fetch_in<NCOMP, OT, LDSR>(sd, scope, s - cfg.o2 - cfg.e2, 0);
    i2e = bl_fetch_in<NCOMP, OT, LDSR>(sd, scope, s - cfg.e2, 3);
    d2e = bl_fetch_in<NCOMP, OT, LDSR>(sd, scope, s - cfg.e2, 4);
  }
  const bool all_null = mx.null() && mo1.null() && i1e.null() && d1e.null() && (NCOMP != 5 || (mo2.null() && i2e.null() && d2e.null()));
  int* const mslot = sd.ring + (s % scope) * MT::INTS;
  int tlo[NCOMP], thi[NCOMP];
  int base = 0, width = 0, data = 0, exists = 0, best = 0;
  sd.end_reached = 0;
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
    if (LDSR && hi - lo + 1 > sd.stride) return -1;   // (uniform: every thread leaves)
    // (a workspace row spans every diagonal of the window: lo >= -plen - 1 and hi <= tlen + 1 by the trimming of the inputs)
    const bool has_i1 = (NCOMP != 1) && (!mo1.null() || !i1e.null());
    const bool has_d1 = (NCOMP != 1) && (!mo1.null() || !d1e.null());
    const bool has_i2 = (NCOMP == 5) && (!mo2.null() || !i2e.null());
    const bool has_d2 = (NCOMP == 5) && (!mo2.null() || !d2e.null());
    base = sd.rbase; width = sd.stride; data = sd.next_data(s, scope); exists = 1;
    const int kmask = sd.kmask;
    const int o_m = data - base;
    const int o_i1 = o_m + width, o_d1 = o_m + 2 * width, o_i2 = o_m + 3 * width, o_d2 = o_m + 4 * width;
    const P wsw = sd.ws;
    const int ak = tlen - plen;
    // reduced below: [0, NCOMP) first in-bounds diagonal per component (min), [NCOMP, 2 NCOMP) last (max), [2 NCOMP] antidiagonal (max),
    // [2 NCOMP + 1] the end component's offset on diagonal tlen - plen (max)
    int r[2 * NCOMP + 2];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) { r[c] = INT_MAX; r[NCOMP + c] = INT_MIN; }
    r[2 * NCOMP] = 0; r[2 * NCOMP + 1] = INT_MIN;
    struct Cell { int mo1l, i1l, mo1r, d1r, mxc, mo2l, i2l, mo2r, d2r; };
    auto load = [&](int k, Cell& c) {
      c.mo1l = mo1.get(ws, k - 1); c.mo1r = mo1.get(ws, k + 1);
      if (NCOMP != 1) { c.i1l = i1e.get(ws, k - 1); c.d1r = d1e.get(ws, k + 1); } else { c.i1l = WFA_OFFSET_NULL; c.d1r = WFA_OFFSET_NULL; }
      c.mxc = (NCOMP == 1 && cfg.metric == 0) ? WFA_OFFSET_NULL : mx.get(ws, k);
      if (NCOMP == 5) { c.mo2l = mo2.get(ws, k - 1); c.mo2r = mo2.get(ws, k + 1); c.i2l = i2e.get(ws, k - 1); c.d2r = d2e.get(ws, k + 1); }
    };
    int best_lane = 0;
    auto inb = [&](int off, int k) -> bool { return (uint32_t)off <= (uint32_t)tlen && (uint32_t)(off - k) <= (uint32_t)plen; };
    // first / last in-bounds diagonal of a component: lanes hold ascending diagonals, so a chunk's are the ends of the compare's lane mask
    auto ends = [&](bool in, int kbase, int c) {
      const unsigned long long m = __ballot(in);
      if (m) { r[c] = min(r[c], kbase + (int)__builtin_ctzll(m)); r[NCOMP + c] = max(r[NCOMP + c], kbase + 63 - (int)__builtin_clzll(m)); }
    };
    auto finish = [&](int k, const Cell& c) {
      const int km = k & kmask;
      const int kbase = __builtin_amdgcn_readfirstlane(k);   // (lanes hold ascending diagonals and leave the loop from the top: the first live lane is lane 0)
      const int ins1 = max(c.mo1l, c.i1l) + 1;
      const int del1 = max(c.mo1r, c.d1r);
      int ins = ins1, del = del1, endv = INT_MIN;
      if (has_i1) { bl_st(wsw, o_i1 + km, ins1); ends(inb(ins1, k), kbase, 1 % NCOMP); if (comp_end == 1) endv = ins1; }
      if (has_d1) { bl_st(wsw, o_d1 + km, del1); ends(inb(del1, k), kbase, 2 % NCOMP); if (comp_end == 2) endv = del1; }
      if (NCOMP == 5) {
        const int ins2 = max(c.mo2l, c.i2l) + 1;
        const int del2 = max(c.mo2r, c.d2r);
        if (has_i2) { bl_st(wsw, o_i2 + km, ins2); ends(inb(ins2, k), kbase, 3 % NCOMP); if (comp_end == 3) endv = ins2; }
        if (has_d2) { bl_st(wsw, o_d2 + km, del2); ends(inb(del2, k), kbase, 4 % NCOMP); if (comp_end == 4) endv = del2; }
        ins = max(ins1, ins2);
        del = max(del1, del2);
      }
      int mv = (NCOMP == 1 && cfg.metric == 0) ? max(del, ins) : max(del, max(c.mxc + 1, ins));
      const bool min_b = inb(mv, k);
      ends(min_b, kbase, 0);
      if (!min_b) {
        mv = WFA_OFFSET_NULL;   // only M is clamped
      } else {
        const int v = mv - k;
        mv += view.run(v, mv, min(plen - v, tlen - mv));
        best_lane = max(best_lane, 2 * mv - k);
      }
      if (comp_end == 0) endv = mv;
      const unsigned long long mak = __ballot(k == ak);
      if (mak) r[2 * NCOMP + 1] = __builtin_amdgcn_readlane(endv, (int)__builtin_ctzll(mak));
      bl_st(wsw, o_m + km, mv);
    };
    // (the workspace form: WFA_BL_FLY chunks of loads in flight per thread — a step is one round trip to the rows)
    constexpr int FLY = LDSR ? 2 : (sizeof(OT) == 4 ? 4 : WFA_BL_FLY);   // (int32 rows = reads beyond 32 kb: wavefronts of tens of thousands of diagonals, 100 kb +16 % with four)
    for (int k = lo + tid; k <= hi; k += FLY * THREADS) {
      Cell c[FLY];
#pragma unroll
      for (int j = 0; j < FLY; ++j) if (j == 0 || k + j * THREADS <= hi) load(k + j * THREADS, c[j]);
#pragma unroll
      for (int j = 0; j < FLY; ++j) if (j == 0 || k + j * THREADS <= hi) finish(k + j * THREADS, c[j]);
    }
    r[2 * NCOMP] = best_lane;
    bl_reduce<THREADS, 2 * NCOMP + 2>(r, (((1u << NCOMP) - 1) << NCOMP) | (3u << (2 * NCOMP)), red, phase, tid, ~(1u << (2 * NCOMP)));   // (all but the antidiagonal are wave values already)
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const bool has = (c == 0) || (c == 1 && has_i1) || (c == 2 && has_d1) || (NCOMP == 5 && c == 3 && has_i2) || (NCOMP == 5 && c == 4 && has_d2);
      if (has && r[c] != INT_MAX) { tlo[c] = r[c]; thi[c] = r[NCOMP + c]; }
    }
    best = r[2 * NCOMP];
    int elo = tlo[0], ehi = thi[0];
#pragma unroll
    for (int c = 1; c < NCOMP; ++c) if (comp_end == c) { elo = tlo[c]; ehi = thi[c]; }
    sd.end_reached = (elo <= ak && ak <= ehi && r[2 * NCOMP + 1] >= tlen) ? 1 : 0;
    sd.cur_exists = 1; sd.cur_lo = tlo[0]; sd.cur_hi = thi[0]; sd.cur_idx0 = o_m;
  }
  __syncthreads();   // every thread has read the inputs' directory records before the slot of score s is overwritten
  if (tid == 0) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) { mslot[MT::LO + c] = tlo[c]; mslot[MT::HI + c] = thi[c]; }
    mslot[MT::BASE] = base; mslot[MT::WIDTH] = width; mslot[MT::DATA] = data; mslot[MT::EXISTS] = exists;
  }
  __syncthreads();
  return best;
}

// R/wavefront_heuristic.c:509-567 on the extended M wavefront of score s (wf-adaptive :257-293, X-drop :297-383), the gap
// wavefronts cut to the same limits (:161-172)
template <int NCOMP, typename OT, int THREADS, bool LDSR>
__device__ __forceinline__ void bl_side_cutoff(BlSide<NCOMP, OT, LDSR>& sd, const WfaDevConfig& cfg, int scope, int s, int plen, int tlen, int* red, int& phase, int tid) {
  typedef Meta<NCOMP> MT;
  if (cfg.heuristic == 0 || !sd.cur_exists || sd.cur_lo > sd.cur_hi) return;
  --sd.steps_wait;
  const int cur_lo = sd.cur_lo, cur_hi = sd.cur_hi;
  int new_lo = cur_lo, new_hi = cur_hi;
  const auto ws = sd.ws;
  const int kmask = sd.kmask;
  if (cfg.heuristic == 1) {
    if (sd.steps_wait <= 0 && (cur_hi - cur_lo + 1) >= cfg.min_wf_len) {
      int dm[1] = {max(plen, tlen)};
      for (int k = cur_lo + tid; k <= cur_hi; k += THREADS) {
        const int off = bl_ld(ws, sd.cur_idx0 + (k & kmask));
        const int d = (off >= 0) ? max(plen - (off - k), tlen - off) : -WFA_OFFSET_NULL;
        dm[0] = min(dm[0], d);
      }
      bl_reduce<THREADS, 1>(dm, 0u, red, phase, tid);
      int lh[2] = {INT_MAX, INT_MIN};
      for (int k = cur_lo + tid; k <= cur_hi; k += THREADS) {
        const int off = bl_ld(ws, sd.cur_idx0 + (k & kmask));
        const int d = (off >= 0) ? max(plen - (off - k), tlen - off) : -WFA_OFFSET_NULL;
        if (d - dm[0] <= cfg.max_dist_thr) { lh[0] = min(lh[0], k); lh[1] = max(lh[1], k); }
      }
      bl_reduce<THREADS, 2>(lh, 2u, red, phase, tid);
      const int ak = tlen - plen;
      const int top_limit = min(ak, cur_hi);
      if (top_limit > cur_lo) new_lo = min(lh[0], top_limit);
      const int bottom_limit = max(ak, new_lo);
      if (bottom_limit < cur_hi) new_hi = max(lh[1], bottom_limit);
      sd.steps_wait = cfg.steps_between;
    }
  } else if (cfg.heuristic == 2) {
    if (sd.steps_wait <= 0) {
      const int g = (cfg.match != 0) ? -cfg.match : -1;  // R/wavefront_heuristic.c:306-307
      int v[3] = {INT_MIN, INT_MAX, INT_MIN};   // cmax, lc, hc
      for (int k = cur_lo + tid; k <= cur_hi; k += THREADS) {
        const int off = bl_ld(ws, sd.cur_idx0 + (k & kmask));
        if (off < 0) continue;
        const int sw = (g * ((off - k) + off) - s) / 2;
        v[0] = max(v[0], sw);
        if (sd.have_max_sw && sd.max_sw - sw < cfg.xdrop) { v[1] = min(v[1], k); v[2] = max(v[2], k); }
      }
      bl_reduce<THREADS, 3>(v, 5u, red, phase, tid);
      if (sd.have_max_sw) {
        if (v[1] == INT_MAX) { new_lo = cur_hi + 1; new_hi = cur_hi; }
        else { new_lo = v[1]; new_hi = v[2]; }
        if (v[0] > sd.max_sw) sd.max_sw = v[0];
      } else {
        sd.max_sw = v[0]; sd.have_max_sw = 1;
      }
      sd.steps_wait = cfg.steps_between;
    }
  }
  if (new_lo != cur_lo || new_hi != cur_hi) {
    sd.cur_lo = new_lo; sd.cur_hi = new_hi;
    __syncthreads();
    if (tid == 0) {
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

// R/wavefront_termination.c:37-113 for wavefront 0 (evaluated only when M[s] exists)
template <int NCOMP, typename OT, bool LDSR>
__device__ __forceinline__ bool bl_side_terminated(const BlSide<NCOMP, OT, LDSR>& sd, int scope, int s, int comp_end, int plen, int tlen) {
  typedef Meta<NCOMP> MT;
  if (!sd.cur_exists) return false;
  const int* m = sd.ring + (s % scope) * MT::INTS;
  const int ak = tlen - plen;
  if (m[MT::LO + comp_end] > ak || ak > m[MT::HI + comp_end]) return false;
  return bl_ld(sd.ws, m[MT::DATA] + comp_end * m[MT::WIDTH] - m[MT::BASE] + (ak & sd.kmask)) >= tlen;
}

// R/wavefront_bialign.c:189-311 over the workgroup: the lowest diagonal of aligner 0 on which the two offsets meet wins.
// hitbuf: NW * 4 ints of LDS.
template <int NCOMP, typename OT, int THREADS, bool LDSR>
__device__ __forceinline__ void bl_breakpoint_cc(const BlSide<NCOMP, OT, LDSR>& s0, const BlSide<NCOMP, OT, LDSR>& s1, const int* m0, const int* m1,
                                                 const WfaDevConfig& cfg, bool forward, int score_0, int score_1, int c,
                                                 int plen, int tlen, BiBreakpoint& bp, int* hitbuf, int tid) {
  typedef Meta<NCOMP> MT;
  const int gap_open = (c == 0) ? 0 : ((c == 1 || c == 2) ? cfg.o1 : cfg.o2);
  const int lo_0 = m0[MT::LO + c], hi_0 = m0[MT::HI + c];
  const int lo_1 = tlen - plen - m1[MT::HI + c], hi_1 = tlen - plen - m1[MT::LO + c];
  if (hi_1 < lo_0 || hi_0 < lo_1) return;
  if (score_0 + score_1 - gap_open >= bp.score) return;
  const int min_hi = min(hi_0, hi_1), max_lo = max(lo_0, lo_1);
  const int i0 = m0[MT::DATA] + c * m0[MT::WIDTH] - m0[MT::BASE], i1 = m1[MT::DATA] + c * m1[MT::WIDTH] - m1[MT::BASE];
  const int kmask = s0.kmask;
  for (int kb = max_lo; kb <= min_hi; kb += THREADS) {
    const int k_0 = kb + tid;
    bool hit = false;
    int o0 = 0, o1 = 0;
    if (k_0 <= min_hi) {
      const int k_1 = tlen - plen - k_0;
      o0 = bl_ld(s0.ws, i0 + (k_0 & kmask)); o1 = bl_ld(s1.ws, i1 + (k_1 & kmask));
      hit = (long long)o0 + o1 >= tlen;
      if (hit && c != 0) {   // interior I/D offsets may lie outside the matrix (they are not clamped): skipped (:222-226,236-240)
        const int kk = forward ? k_0 : k_1, oo = forward ? o0 : o1;
        if (oo - kk > plen || oo > tlen) hit = false;
      }
    }
    const unsigned long long bm = __ballot(hit);
    int fk0 = INT_MAX, fo0 = 0, fo1 = 0;
    if (bm) {
      const int L = __builtin_ctzll(bm);
      fk0 = kb + (tid & ~63) + L;
      fo0 = __builtin_amdgcn_readlane(o0, L); fo1 = __builtin_amdgcn_readlane(o1, L);
    }
    if (THREADS > 64) {
      constexpr int NW = THREADS / 64;
      if ((tid & 63) == 0) { hitbuf[(tid >> 6) * 4] = fk0; hitbuf[(tid >> 6) * 4 + 1] = fo0; hitbuf[(tid >> 6) * 4 + 2] = fo1; }
      __syncthreads();
      fk0 = INT_MAX;
#pragma unroll
      for (int w = 0; w < NW; ++w) {   // (waves hold ascending diagonals: the first wave with a hit holds the lowest)
        if (fk0 == INT_MAX && hitbuf[w * 4] != INT_MAX) { fk0 = hitbuf[w * 4]; fo0 = hitbuf[w * 4 + 1]; fo1 = hitbuf[w * 4 + 2]; }
      }
      __syncthreads();
    }
    if (fk0 != INT_MAX) {
      const int fk1 = tlen - plen - fk0;
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
template <int NCOMP, typename OT, int THREADS, bool LDSR>
__device__ __forceinline__ void bl_overlap(const BlSide<NCOMP, OT, LDSR>& s0, const BlSide<NCOMP, OT, LDSR>& s1, const WfaDevConfig& cfg, int scope,
                                           int score_0, int score_1, bool forward, int plen, int tlen, BiBreakpoint& bp, int* hitbuf, int tid) {
  typedef Meta<NCOMP> MT;
  if (s0.g > 1 && score_0 % s0.g != 0) return;   // (null steps leave no record)
  const int* m0 = s0.ring + (score_0 % scope) * MT::INTS;
  if (!m0[MT::EXISTS]) return;
  for (int i = 0; i < scope; ++i) {
    const int score_i = score_1 - i;
    if (score_i < 0) break;
    if (s0.g > 1 && score_i % s0.g != 0) continue;
    const int* m1 = s1.ring + (score_i % scope) * MT::INTS;
    if (NCOMP == 5) {
      if (score_0 + score_i - cfg.o2 >= bp.score) continue;
      bl_breakpoint_cc<NCOMP, OT, THREADS, LDSR>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 4, plen, tlen, bp, hitbuf, tid);
      bl_breakpoint_cc<NCOMP, OT, THREADS, LDSR>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 3, plen, tlen, bp, hitbuf, tid);
    }
    if (NCOMP >= 3) {
      if (score_0 + score_i - cfg.o1 >= bp.score) continue;
      bl_breakpoint_cc<NCOMP, OT, THREADS, LDSR>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 2, plen, tlen, bp, hitbuf, tid);
      bl_breakpoint_cc<NCOMP, OT, THREADS, LDSR>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 1, plen, tlen, bp, hitbuf, tid);
    }
    if (score_0 + score_i >= bp.score) continue;
    if (m1[MT::EXISTS]) bl_breakpoint_cc<NCOMP, OT, THREADS, LDSR>(s0, s1, m0, m1, cfg, forward, score_0, score_i, 0, plen, tlen, bp, hitbuf, tid);
  }
}

// ---- queues and leaves ----
__device__ __forceinline__ void bl_flag(const BlArgs& a, int pair, int f) { atomicOr(a.flags + pair, f); }

// one thread: a leaf of `pair` (ops [begin, begin + n) of its region, which spans [start, start + region))
__device__ __forceinline__ void bl_add_leaf(const BlArgs& a, int pair, int start, int region, int begin, int n) {
  const uint32_t li = atomicAdd(a.cnt + WFA_BL_MAX_LEVELS + 1, 1u);
  if (li >= a.leafcap) { bl_flag(a, pair, WFA_BL_FLAG_REDO); return; }
  BlLeaf lf; lf.start = start; lf.region = region; lf.begin = begin; lf.n = n; lf.pad0 = lf.pad1 = lf.pad2 = 0;
  lf.next = atomicExch(a.head + pair, (int)li);
  a.leaves[li] = lf;
}
__device__ __forceinline__ void bl_fail(const BlArgs& a, int pair, int start, int status) {
  atomicMin(a.failkey + pair, ((unsigned long long)(uint32_t)start << 32) | (uint32_t)status);
}

// a window on its way: written out at once when one of its sequences is empty (R/wavefront_bialign.c:590-606), else queued for the
// base kernel (score <= 250, :607-612) or the next level.  Called by every thread of the workgroup.
template <int THREADS>
__device__ __forceinline__ void bl_route(const BlArgs& a, const BlWindow& w, int next_level, int tid) {
  const int plen = w.pend - w.pbeg, tlen = w.tend - w.tbeg;
  if (plen == 0 && tlen == 0) return;
  if (tlen == 0 || plen == 0) {
    uint8_t* const out = a.k.cigar_ops + a.k.cigar_off[w.pair] + w.pbeg + w.tbeg;
    const int n = plen + tlen;
    const uint8_t op = (tlen == 0) ? 'D' : 'I';
    for (int i = tid; i < n; i += THREADS) out[i] = op;
    if (tid == 0) bl_add_leaf(a, w.pair, w.pbeg + w.tbeg, n, w.pbeg + w.tbeg, n);
    return;
  }
  if (tid != 0) return;
  if (w.score_remaining <= WFA_BI_FALLBACK_MIN_SCORE || (w.flags & (1 << 10))) {
    const uint32_t qi = atomicAdd(a.cnt + WFA_BL_MAX_LEVELS, 1u);
    if (qi >= a.qbcap) { bl_flag(a, w.pair, WFA_BL_FLAG_REDO); return; }
    a.qb[qi] = w;
  } else {
    if (next_level >= WFA_BL_MAX_LEVELS) { bl_flag(a, w.pair, WFA_BL_FLAG_REDO); return; }
    const uint32_t qi = atomicAdd(a.cnt + next_level, 1u);
    if (qi >= a.qcap) { bl_flag(a, w.pair, WFA_BL_FLAG_REDO); return; }
    a.q[next_level & 1][qi] = w;
  }
}

template <int UNIT>   // (a template so that every translation unit including this header may hold it)
__global__ void __launch_bounds__(256) bl_seed_kernel(const BlArgs a) {
  const uint32_t wi = blockIdx.x * 256 + threadIdx.x;
  if (wi >= a.k.nwork) return;
  const uint32_t pair = a.k.worklist ? a.k.worklist[wi] : wi;
  const WfaPairMeta pm = a.k.meta[pair];
  a.head[pair] = -1; a.flags[pair] = 0; a.top[pair] = INT_MIN; a.failkey[pair] = ~0ull;
  BlWindow w;
  w.pair = (int)pair; w.pbeg = 0; w.pend = pm.plen; w.tbeg = 0; w.tend = pm.tlen;
  w.flags = 0 | (0 << 4) | (1 << 8) | ((a.k.cfg.endsfree ? 1 : 0) << 9);
  w.score_remaining = (max(pm.plen, pm.tlen) <= WFA_BI_FALLBACK_MIN_LENGTH) ? 0 : INT_MAX;
  w.pad = 0;
  // (bl_route's one-thread part: the seed kernel runs one thread per pair)
  if (pm.plen == 0 && pm.tlen == 0) return;
  if (pm.plen == 0 || pm.tlen == 0) {
    uint8_t* const out = a.k.cigar_ops + a.k.cigar_off[pair];
    const int n = pm.plen + pm.tlen;
    for (int i = 0; i < n; ++i) out[i] = (pm.tlen == 0) ? 'D' : 'I';
    bl_add_leaf(a, (int)pair, 0, n, 0, n);
    return;
  }
  if (w.score_remaining == 0) {
    const uint32_t qi = atomicAdd(a.cnt + WFA_BL_MAX_LEVELS, 1u);
    if (qi >= a.qbcap) { bl_flag(a, (int)pair, WFA_BL_FLAG_REDO); return; }
    a.qb[qi] = w;
  } else {
    const uint32_t qi = atomicAdd(a.cnt + 0, 1u);
    if (qi >= a.qcap) { bl_flag(a, (int)pair, WFA_BL_FLAG_REDO); return; }
    a.q[0][qi] = w;
  }
}

#ifndef WFA_BL_WAVES
#define WFA_BL_WAVES 5
#endif
template <int NCOMP, bool PACKED, typename OT, int THREADS, bool LDSR, bool SEQL>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 1024 ? 4 : (THREADS == 256 && !LDSR ? WFA_BL_WAVES : 5))))   // (<= 96 registers: five waves per SIMD)
bl_split_kernel(const BlArgs a) {
  typedef Meta<NCOMP> MT;
  typedef typename BlPtr<OT, LDSR>::type P;
  typedef typename BlPtr<const uint32_t, SEQL>::type WP;
  typedef BlView<PACKED, SEQL> View;
  static_assert(!LDSR || SEQL, "rows in LDS: the sequences too");
  static_assert(!SEQL || PACKED, "sequences in LDS: 2-bit pairs");
  extern __shared__ int smem[];
  const WfaDevConfig& cfg = a.k.cfg;
  const int scope = cfg.scope;
  const int tid = threadIdx.x;
  int* const ring_f = smem;
  int* const ring_r = ring_f + scope * MT::INTS;
  int* const red = ring_r + scope * MT::INTS;          // three buffers of 32 (bl_reduce)
  int* const hitbuf = red + 96;                        // (THREADS / 64) * 4, then the window index
  int* const next_wi = hitbuf + (THREADS / 64) * 4;
  // LDSR: the two sequences' words and the rows of both aligners behind it
  int* const lds_seq = next_wi + 8;
  const int row_elems = a.lds_slots * NCOMP * a.lds_w;
  OT* const wsb = LDSR ? nullptr : reinterpret_cast<OT*>(reinterpret_cast<char*>(a.rings) + (long long)blockIdx.x * a.slice_bytes);
  const int cnt_word = a.from_wide ? 128 + a.level : a.level;
  const uint32_t nwork = min(a.cnt[cnt_word], a.from_wide ? a.qwcap : a.qcap);
  const BlWindow* const q = a.from_wide ? a.qw : a.q[a.level & 1];
  const long long max_steps = cfg.max_steps;
  int phase = 0;
  bl_reduce_init(red, tid);
  const int lattice_g = bl_lattice_gcd<NCOMP>(cfg);

  for (;;) {
    // windows differ in cost by orders of magnitude: taken one at a time from the level's counter
    __syncthreads();
    if (tid == 0) next_wi[0] = (int)atomicAdd(a.cnt + cnt_word + 64, 1u);
    __syncthreads();
    const uint32_t wi = (uint32_t)next_wi[0];
    if (wi >= nwork) break;
    const BlWindow w = q[wi];
    const uint32_t pair = (uint32_t)w.pair;
    const WfaPairMeta pm = a.k.meta[pair];
    const int pbeg = w.pbeg, pend = w.pend, tbeg = w.tbeg, tend = w.tend;
    const int comp_begin = w.flags & 15, comp_end = (w.flags >> 4) & 15;
    const bool level0 = (w.flags >> 8) & 1;
    const int plen = pend - pbeg, tlen = tend - tbeg;
    bool overflow = false;
    View view;
    view.wildcard = cfg.wildcard;
    view.pb = nullptr; view.tb = nullptr; view.pw = nullptr; view.tw = nullptr;
    if (SEQL) {
      // the words the window's probes can touch (one word before its first base: the backward probes; two behind its last)
      const uint32_t* gP = a.k.words + pm.p_woff; const uint32_t* gT = a.k.words + pm.t_woff;
      const int nwp = (pm.plen + 15) >> 4, nwt = (pm.tlen + 15) >> 4;
      const int p0 = max(0, (pbeg >> 4) - 1), p1 = ((pend + 15) >> 4) + 2;
      const int t0 = max(0, (tbeg >> 4) - 1), t1 = ((tend + 15) >> 4) + 2;
      if (p1 - p0 > a.lds_seq_words || t1 - t0 > a.lds_seq_words) overflow = true;
      else {
        uint32_t* sP = reinterpret_cast<uint32_t*>(lds_seq); uint32_t* sT = sP + a.lds_seq_words;
        for (int i = p0 + tid; i < p1; i += THREADS) sP[i - p0] = (i < nwp) ? gP[i] : 0u;
        for (int i = t0 + tid; i < t1; i += THREADS) sT[i - t0] = (i < nwt) ? gT[i] : 0u;
        view.pw = (WP)sP - p0; view.tw = (WP)sT - t0;
      }
    } else if (PACKED) {
      view.pw = (WP)(a.k.words + pm.p_woff); view.tw = (WP)(a.k.words + pm.t_woff);
    } else {
      view.pb = a.k.bytes + a.k.p_boff[pair]; view.tb = a.k.bytes + a.k.t_boff[pair];
    }
    view.pbeg = pbeg; view.pend = pend; view.tbeg = tbeg; view.tend = tend; view.reverse = false;
    View rview = view; rview.reverse = true;
    BlSide<NCOMP, OT, LDSR> F, R;
    F.ring = ring_f; R.ring = ring_r;
    F.lds_slots = R.lds_slots = a.lds_slots;
    F.g = R.g = lattice_g;
    if (LDSR) {
      OT* rows = reinterpret_cast<OT*>(lds_seq + 2 * a.lds_seq_words);
      F.ws = (P)rows; R.ws = (P)(rows + row_elems);
      F.stride = R.stride = a.lds_w;
    } else {
      const int stride = min(a.ring_stride, (plen + tlen + 3 + 1) & ~1);
      F.ws = (P)wsb; R.ws = (P)(wsb + a.ring_elems);
      F.stride = R.stride = stride;
    }
    BiBreakpoint bp;
    bp.score = INT_MAX; bp.score_forward = 0; bp.score_reverse = 0; bp.k_forward = 0; bp.k_reverse = 0;
    bp.offset_forward = 0; bp.offset_reverse = 0; bp.component = 0;
    // ---------------- R/wavefront_bialign.c:411-519 (wavefront_bialign_find_breakpoint) ----------------
    int st = WFA_BI_OK, reached = 0;
    bool quit = false;
    if (!overflow) {
      bl_side_init<NCOMP, OT, LDSR>(F, scope, comp_begin, plen, tlen, tid);
      bl_side_init<NCOMP, OT, LDSR>(R, scope, comp_end, plen, tlen, tid);
      F.steps_wait = R.steps_wait = cfg.steps_between;   // (R/wavefront_heuristic.c:114-121)
      const int max_antidiagonal = plen + tlen - 1;
      int score_f = 0, score_r = 0;
      // what follows the extension of a wavefront: end test, cut-off; true when that aligner is done
      auto after = [&](BlSide<NCOMP, OT, LDSR>& sd, int s, int cend, int best, int* max_ak) -> bool {
        if (best < 0) { overflow = true; return true; }
        if (!sd.cur_exists) {
          *max_ak = 0;
          if (sd.null_steps > scope) { st = WFA_BI_END_UNREACHABLE; reached = s; return true; }
          return false;
        }
        const bool ended = (s == 0) ? bl_side_terminated<NCOMP, OT, LDSR>(sd, scope, s, cend, plen, tlen) : (sd.end_reached != 0);
        if (ended) { st = WFA_BI_END_REACHED; reached = s; *max_ak = 0; return true; }
        bl_side_cutoff<NCOMP, OT, THREADS, LDSR>(sd, cfg, scope, s, plen, tlen, red, phase, tid);
        *max_ak = best;
        return false;
      };
      auto stepF = [&]() -> int { return bl_side_step<NCOMP, OT, THREADS, LDSR, View>(F, view, cfg, scope, score_f, comp_end, plen, tlen, red, phase, tid); };
      auto stepR = [&]() -> int { return bl_side_step<NCOMP, OT, THREADS, LDSR, View>(R, rview, cfg, scope, score_r, comp_begin, plen, tlen, red, phase, tid); };
      int f_max_ak = 0, r_max_ak = 0, max_ak = 0;
      quit = after(F, 0, comp_end, bl_side_extend0<NCOMP, OT, THREADS, LDSR, View>(F, view, plen, tlen, red, phase, tid), &f_max_ak);
      if (!quit) quit = after(R, 0, comp_begin, bl_side_extend0<NCOMP, OT, THREADS, LDSR, View>(R, rview, plen, tlen, red, phase, tid), &r_max_ak);
      bool last_forward = false;
      while (!quit) {
        if (f_max_ak + r_max_ak >= max_antidiagonal) break;
        ++score_f;
        quit = after(F, score_f, comp_end, stepF(), &max_ak);
        if (f_max_ak < max_ak) f_max_ak = max_ak;
        last_forward = true;
        if (quit) break;
        if (f_max_ak + r_max_ak >= max_antidiagonal) break;
        ++score_r;
        quit = after(R, score_r, comp_begin, stepR(), &max_ak);
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
            bl_overlap<NCOMP, OT, THREADS, LDSR>(F, R, cfg, scope, score_f, score_r, true, plen, tlen, bp, hitbuf, tid);
            ++score_r;
            if (after(R, score_r, comp_begin, stepR(), &max_ak)) { quit = true; break; }
          }
          const int min_score_forward = (score_f > scope - 1) ? score_f - (scope - 1) : 0;
          if (min_score_forward + score_r - gap_opening >= bp.score) break;
          bl_overlap<NCOMP, OT, THREADS, LDSR>(R, F, cfg, scope, score_r, score_f, false, plen, tlen, bp, hitbuf, tid);
          ++score_f;
          if (after(F, score_f, comp_end, stepF(), &max_ak)) { quit = true; break; }
          if ((long long)score_r + score_f >= max_steps) { st = WFA_STATUS_MAX_STEPS_REACHED; quit = true; break; }
          last_forward = true;
        }
      }
    }
    if (overflow) {
      // the LDS rows (or the sequence buffers) do not hold this window: the workspace form of the same level takes it
      if (tid == 0) {
        const uint32_t qi = atomicAdd(a.cnt + 128 + a.level, 1u);
        if (qi >= a.qwcap) bl_flag(a, (int)pair, WFA_BL_FLAG_REDO); else a.qw[qi] = w;
      }
      continue;
    }
    if (quit) {
      // R/wavefront_bialign.c:520-548 (wavefront_bialign_find_breakpoint_exception)
      if (st == WFA_BI_END_REACHED && reached <= WFA_BI_RECOVERY_MIN_SCORE) {
        BlWindow wb = w; wb.flags |= (1 << 10);
        bl_route<THREADS>(a, wb, a.level + 1, tid);
      } else if (tid == 0) {
        bl_fail(a, (int)pair, pbeg + tbeg, (st == WFA_STATUS_MAX_STEPS_REACHED) ? WFA_STATUS_MAX_STEPS_REACHED : WFA_STATUS_UNATTAINABLE);
      }
      continue;
    }
    // ---------------- breakpoint found: the two halves (R/wavefront_bialign.c:614-650) ----------------
    const int bh = bp.offset_forward, bv = bp.offset_forward - bp.k_forward;
    if (level0 && tid == 0) a.top[pair] = bp.score;
    BlWindow w0, w1;
    w0.pair = w1.pair = (int)pair; w0.pad = w1.pad = 0;
    w0.pbeg = pbeg; w0.pend = pbeg + bv; w0.tbeg = tbeg; w0.tend = tbeg + bh;
    w0.flags = comp_begin | (bp.component << 4); w0.score_remaining = bp.score_forward;
    w1.pbeg = pbeg + bv; w1.pend = pend; w1.tbeg = tbeg + bh; w1.tend = tend;
    w1.flags = bp.component | (comp_end << 4); w1.score_remaining = bp.score_reverse;
    bl_route<THREADS>(a, w0, a.level + 1, tid);
    bl_route<THREADS>(a, w1, a.level + 1, tid);
  }
}

// One wave per base window: wfa_biwfa_kernel's base case (R/wavefront_bialign.c:155-188) with its int32 history.
template <int NCOMP, bool PACKED>
__global__ void __launch_bounds__(64)
bl_base_kernel(const BlArgs a) {
  typedef Meta<NCOMP> MT;
  extern __shared__ int smem[];
  const WfaDevConfig& cfg = a.k.cfg;
  const int scope = cfg.scope;
  const int lane = threadIdx.x;
  int* const ring_b = smem;
  int* const wsb = reinterpret_cast<int*>(reinterpret_cast<char*>(a.rings) + (long long)blockIdx.x * a.slice_bytes);
  const uint32_t nwork = min(a.cnt[WFA_BL_MAX_LEVELS], a.qbcap);
  const long long max_steps = cfg.max_steps;
  int* const next_wi = ring_b + scope * MT::INTS;
  const int lattice_g = bl_lattice_gcd<NCOMP>(cfg);
  for (;;) {
    __syncthreads();
    if (lane == 0) next_wi[0] = (int)atomicAdd(a.cnt + 64 + WFA_BL_MAX_LEVELS, 1u);
    __syncthreads();
    const uint32_t wi = (uint32_t)next_wi[0];
    if (wi >= nwork) break;
    const BlWindow w = a.qb[wi];
    const uint32_t pair = (uint32_t)w.pair;
    const WfaPairMeta pm = a.k.meta[pair];
    BiView<PACKED> view;
    view.wildcard = cfg.wildcard;
    if (PACKED) { view.pw = a.k.words + pm.p_woff; view.tw = a.k.words + pm.t_woff; view.pb = nullptr; view.tb = nullptr; }
    else { view.pb = a.k.bytes + a.k.p_boff[pair]; view.tb = a.k.bytes + a.k.t_boff[pair]; view.pw = nullptr; view.tw = nullptr; }
    const int comp_begin = w.flags & 15, comp_end = (w.flags >> 4) & 15;
    const bool ef_form = (w.flags >> 9) & 1;
    const int plen = w.pend - w.pbeg, tlen = w.tend - w.tbeg;
    view.pbeg = w.pbeg; view.pend = w.pend; view.tbeg = w.tbeg; view.tend = w.tend; view.reverse = false;
    uint8_t* const out = a.k.cigar_ops + a.k.cigar_off[pair];
    BiSide<NCOMP> B;
    B.ring = ring_b; B.ws = wsb; B.slots = WFA_BI_BASE_SLOTS;
    B.dir = wsb + a.base_ints - (long long)WFA_BI_BASE_SLOTS * MT::INTS;
    B.stride = min(a.base_stride, plen + tlen + 3);
    bi_side_init<NCOMP>(B, scope, comp_begin, plen, tlen, lane);
    int s = 0, end_k = 0, end_off = 0;
    bool reached_end = false, fail = false, hand_on = false;
    while (true) {
      if (!B.cur_exists) {
        if (B.null_steps > scope) { fail = true; break; }
      } else if (ef_form) {
        bi_side_extend<NCOMP, PACKED>(B, view, plen, tlen, lane);
        const int ak = tlen - plen;
        if (B.cur_lo <= ak && ak <= B.cur_hi && B.ws[B.cur_idx0 + ak] >= tlen) { reached_end = true; end_k = ak; end_off = tlen; break; }
      } else {
        bi_side_extend<NCOMP, PACKED>(B, view, plen, tlen, lane);
        if (bi_side_terminated<NCOMP>(B, scope, s, comp_end, plen, tlen)) { reached_end = true; end_k = tlen - plen; end_off = tlen; break; }
      }
      ++s;
      if (s >= max_steps) { fail = true; break; }
      if (s >= WFA_BI_BASE_SLOTS - 1) { hand_on = true; break; }
      if (lattice_g > 1 && s % lattice_g != 0) {   // a null step (no sum of the penalties): registers only, its records are never read
        ++B.null_steps; B.cur_exists = 0; B.cur_lo = 1; B.cur_hi = -1; B.cur_idx0 = 0;
        continue;
      }
      if (!bi_side_compute<NCOMP>(B, cfg, scope, s, plen, tlen, lane)) { hand_on = true; break; }
    }
    const int start = w.pbeg + w.tbeg;
    if (hand_on) { if (lane == 0) bl_flag(a, (int)pair, WFA_BL_FLAG_HANDON); __syncthreads(); continue; }
    if (fail || !reached_end) { if (lane == 0) bl_fail(a, (int)pair, start, WFA_STATUS_UNATTAINABLE); __syncthreads(); continue; }
    __syncthreads();
    if (lane == 0) {
      const long long end_pos = (long long)start + plen + tlen;
      const long long begin = bi_backtrace<NCOMP>(B, cfg, plen, tlen, s, end_k, end_off, comp_end, out, end_pos);
      bl_add_leaf(a, (int)pair, start, plen + tlen, (int)begin, (int)(end_pos - begin));
    }
    __syncthreads();
  }
}

// One wave per pair: leaves sorted by start, op strings moved together, results written.
template <int UNIT>
__global__ void __launch_bounds__(64)
bl_finish_kernel(const BlArgs a) {
  __shared__ int l_start[WFA_BL_LEAF_LDS], l_region[WFA_BL_LEAF_LDS], l_begin[WFA_BL_LEAF_LDS], l_n[WFA_BL_LEAF_LDS];
  __shared__ int order[WFA_BL_LEAF_LDS];
  __shared__ int s_count;
  const int lane = threadIdx.x;
  for (uint32_t wi = blockIdx.x; wi < a.k.nwork; wi += gridDim.x) {
    const uint32_t pair = a.k.worklist ? a.k.worklist[wi] : wi;
    const WfaPairMeta pm = a.k.meta[pair];
    int flags = a.flags[pair];
    __syncthreads();
    if (lane == 0) {
      int cnt = 0;
      if (!(flags & WFA_BL_FLAG_REDO)) {
        for (int li = a.head[pair]; li >= 0; ) {
          if (cnt >= WFA_BL_LEAF_LDS) { cnt = -1; break; }
          const BlLeaf lf = a.leaves[li];
          l_start[cnt] = lf.start; l_region[cnt] = lf.region; l_begin[cnt] = lf.begin; l_n[cnt] = lf.n;
          ++cnt; li = lf.next;
        }
      }
      s_count = cnt;
    }
    __syncthreads();
    const int cnt = s_count;
    if (cnt < 0) flags |= WFA_BL_FLAG_REDO;
    const unsigned long long fk = a.failkey[pair];
    const bool failed = fk != ~0ull;
    const int fail_start = failed ? (int)(fk >> 32) : INT_MAX;
    const int fail_status = (int)(uint32_t)fk;
    uint8_t* const out = a.k.cigar_ops + a.k.cigar_off[pair];
    long long out_len = 0;
    if (!(flags & (WFA_BL_FLAG_REDO | WFA_BL_FLAG_HANDON))) {
      // rank of every leaf by start (starts are distinct: the regions of leaves are non-empty and disjoint)
      for (int i = lane; i < cnt; i += 64) {
        const int si = l_start[i];
        int rank = 0;
        for (int j = 0; j < cnt; ++j) rank += (l_start[j] < si) ? 1 : 0;
        order[rank] = i;
      }
      __syncthreads();
      int expect = 0;
      bool tiled = true;
      for (int r = 0; r < cnt; ++r) {
        const int i = order[r];
        const int st = l_start[i];
        if (st >= fail_start) break;
        if (st != expect) { tiled = false; break; }
        expect += l_region[i];
        const int src = l_begin[i], n = l_n[i];
        if (src != out_len) {
          for (int i0 = 0; i0 < n; i0 += 256) {
            uint8_t c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int x = i0 + lane * 4 + j; c[j] = (x < n) ? out[src + x] : 0; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int x = i0 + lane * 4 + j; if (x < n) out[out_len + x] = c[j]; }
            __syncthreads();
          }
        }
        out_len += n;
      }
      if (tiled && expect != (failed ? fail_start : pm.plen + pm.tlen)) tiled = false;
      if (!tiled) flags |= WFA_BL_FLAG_REDO;
    }
    if (lane == 0) {
      if (flags & WFA_BL_FLAG_HANDON) {
        a.k.score[pair] = INT_MIN; a.k.cigar_begin[pair] = a.k.cigar_off[pair]; a.k.cigar_len[pair] = 0;
        if (a.k.fb_list) { a.k.status[pair] = WFA_INTERNAL_FALLBACK; a.k.fb_list[atomicAdd(a.k.fb_count, 1u)] = pair; }
        else a.k.status[pair] = WFA_STATUS_UNATTAINABLE;
      } else if (flags & WFA_BL_FLAG_REDO) {
        a.redo_list[atomicAdd(a.cnt + WFA_BL_MAX_LEVELS + 2, 1u)] = pair;   // (wfa_biwfa_kernel writes the results)
      } else {
        const int top = a.top[pair];
        a.k.score[pair] = (!failed && top != INT_MIN) ? classic_score(a.k.cfg, pm.plen, pm.tlen, top) : INT_MIN;
        a.k.status[pair] = failed ? fail_status : 0;
        a.k.cigar_begin[pair] = a.k.cigar_off[pair];
        a.k.cigar_len[pair] = (int)out_len;
      }
    }
    __syncthreads();
  }
}

// host entry points (csrc/k_bilevel.hip, one translation unit per component count)
int launch_bl_split_c1(bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_split_c3(bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_split_c5(bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_split_lds_c1(int threads, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_split_lds_c3(int threads, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_base_c1(bool packed, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_base_c3(bool packed, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_base_c5(bool packed, const BlArgs& a, int grid, size_t smem, hipStream_t stream);
int launch_bl_seed(const BlArgs& a, hipStream_t stream);
int launch_bl_finish(const BlArgs& a, int grid, hipStream_t stream);

template <int NCOMP>
inline int launch_bl_split_ncomp(bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, size_t smem, hipStream_t stream) {
#define WFA_BL_LAUNCH(P, OT, T, SQ) hipLaunchKernelGGL((bl_split_kernel<NCOMP, P, OT, T, false, SQ>), dim3(grid), dim3(T), smem, stream, a)
  if (threads == 1024 && packed) {   // (windows tens of thousands of diagonals wide: 100 kb reads' top levels; 2-bit pairs)
    if (seql) { if (i16) WFA_BL_LAUNCH(true, short, 1024, true); else WFA_BL_LAUNCH(true, int, 1024, true); }
    else { if (i16) WFA_BL_LAUNCH(true, short, 1024, false); else WFA_BL_LAUNCH(true, int, 1024, false); }
  } else if (threads >= 256) {
    if (packed && seql) { if (i16) WFA_BL_LAUNCH(true, short, 256, true); else WFA_BL_LAUNCH(true, int, 256, true); }
    else if (packed) { if (i16) WFA_BL_LAUNCH(true, short, 256, false); else WFA_BL_LAUNCH(true, int, 256, false); }
    else { if (i16) WFA_BL_LAUNCH(false, short, 256, false); else WFA_BL_LAUNCH(false, int, 256, false); }
  } else {
    if (packed && seql) { if (i16) WFA_BL_LAUNCH(true, short, 64, true); else WFA_BL_LAUNCH(true, int, 64, true); }
    else if (packed) { if (i16) WFA_BL_LAUNCH(true, short, 64, false); else WFA_BL_LAUNCH(true, int, 64, false); }
    else { if (i16) WFA_BL_LAUNCH(false, short, 64, false); else WFA_BL_LAUNCH(false, int, 64, false); }
  }
#undef WFA_BL_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// the LDS form: 2-bit pairs, int16 rows; 64 / 256 / 1024 threads
template <int NCOMP>
inline int launch_bl_split_lds_ncomp(int threads, const BlArgs& a, int grid, size_t smem, hipStream_t stream) {
#define WFA_BL_LAUNCH(T) do { \
    static bool attr_set = false; \
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bl_split_kernel<NCOMP, true, short, T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL((bl_split_kernel<NCOMP, true, short, T, true, true>), dim3(grid), dim3(T), smem, stream, a); } while (0)
  if (threads == 1024) WFA_BL_LAUNCH(1024);
  else if (threads == 256) WFA_BL_LAUNCH(256);
  else WFA_BL_LAUNCH(64);
#undef WFA_BL_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int NCOMP>
inline int launch_bl_base_ncomp(bool packed, const BlArgs& a, int grid, size_t smem, hipStream_t stream) {
  if (packed) hipLaunchKernelGGL((bl_base_kernel<NCOMP, true>), dim3(grid), dim3(64), smem, stream, a);
  else hipLaunchKernelGGL((bl_base_kernel<NCOMP, false>), dim3(grid), dim3(64), smem, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

inline size_t bl_split_smem(int ncomp, int scope, int threads) {
  return ((size_t)2 * scope * (2 * ncomp + 4) + 96 + (size_t)(threads / 64) * 4 + 16) * sizeof(int);
}
// the LDS form: + two sequence buffers + the rows of both aligners (int16)
inline size_t bl_split_lds_smem(int ncomp, int scope, int threads, int w, int slots, int seq_words) {
  return bl_split_smem(ncomp, scope, threads) + (size_t)2 * seq_words * 4 + (size_t)2 * slots * ncomp * w * 2;
}
inline size_t bl_base_smem(int ncomp, int scope) { return ((size_t)scope * (2 * ncomp + 4) + 8) * sizeof(int); }

inline int launch_bl_split_any(int ncomp, bool packed, bool i16, int threads, bool seql, const BlArgs& a, int grid, hipStream_t stream) {
  const size_t smem = bl_split_smem(ncomp, a.k.cfg.scope, threads) + (seql ? (size_t)2 * a.lds_seq_words * 4 : 0);
  if (ncomp == 1) return launch_bl_split_c1(packed, i16, threads, seql, a, grid, smem, stream);
  if (ncomp == 3) return launch_bl_split_c3(packed, i16, threads, seql, a, grid, smem, stream);
  return launch_bl_split_c5(packed, i16, threads, seql, a, grid, smem, stream);
}
inline int launch_bl_split_lds_any(int ncomp, int threads, const BlArgs& a, int grid, hipStream_t stream) {
  const size_t smem = bl_split_lds_smem(ncomp, a.k.cfg.scope, threads, a.lds_w, a.lds_slots, a.lds_seq_words);
  if (ncomp == 1) return launch_bl_split_lds_c1(threads, a, grid, smem, stream);
  return launch_bl_split_lds_c3(threads, a, grid, smem, stream);
}
inline int launch_bl_base_any(int ncomp, bool packed, const BlArgs& a, int grid, hipStream_t stream) {
  const size_t smem = bl_base_smem(ncomp, a.k.cfg.scope);
  if (ncomp == 1) return launch_bl_base_c1(packed, a, grid, smem, stream);
  if (ncomp == 3) return launch_bl_base_c3(packed, a, grid, smem, stream);
  return launch_bl_base_c5(packed, a, grid, smem, stream);
}

}  // namespace wfa
