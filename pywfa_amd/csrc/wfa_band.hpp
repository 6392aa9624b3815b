// wfa_band.hpp — wave-resident banded kernel: one alignment per 64-lane workgroup, the wavefront ring
// in VGPRs over a SLIDING window of W = 64*NCH diagonals (diagonal k = B + 64*c + lane for chunk c),
// sequences staged in LDS when they fit, wf-adaptive cut-off with ballots, and — for scope=full — the
// per-score wavefronts streamed to an HBM history (fixed-stride records, coalesced stores) followed by
// an on-device backtrace.  It is the long-read / full-CIGAR companion of the short-read kernels (wfa_lane.hpp, wfa_seg.hpp) and follows the
// same rules (R = /root/reference/pywfa/WFA2_lib/wavefront):
//   compute-next        R/wavefront_compute_affine.c:44-86   (only M clamped; I/D ends trimmed, :571-605)
//   extend/termination  R/wavefront_extend_kernels.c:64-110, R/wavefront_termination.c:37-61
//   wf-adaptive         R/wavefront_heuristic.c:176-293 + equate :161-172, dispatcher :509-567
//   backtrace           R/wavefront_backtrace.c:49-101,320-529
// "Outside a wavefront's [lo,hi] reads NULL" is represented by NULL lanes; the heuristic's cut and the
// I/D equate NULL the lanes they drop, so the stored history needs no per-component limits.
// Scope: gap-affine, match = 0, penalties in the ratio x : o+e : e = 2 : 4 : 1 (pywfa's 4/6/2), no
// heuristic or wf-adaptive, end-to-end (or ends-free with all free ends 0), unlimited max_steps.
// Whatever does not fit (window, history capacity, dead wavefront) is appended to the fallback list and
// finished by the general kernel with identical results.
#pragma once
#include "wfa_rtc_compat.hpp"
#ifndef __HIPCC_RTC__
#include <string>
#include "wfa_rtc.hpp"
#endif
#include "wfa_hip.h"
#include "wfa_common.hpp"
#include "wfa_general.hpp"

namespace wfa {

template <int N> struct band_int { static constexpr int value = N; };   // (a compile-time chunk count passed to the step lambdas)

// penalty shapes (x, o + e, e) / gcd the banded kernel is instantiated for: pywfa's 4/6/2 and the presets 4/4/2, 4/6/1, 3/4/1
#define WFA_BAND_SHAPES(F) F(0, 2, 4, 1) F(1, 2, 3, 1) F(2, 4, 7, 1) F(3, 3, 5, 1)

struct BandArgs {
  const uint32_t* words;
  const WfaPairMeta* meta;
  const uint32_t* worklist;  // nullptr = identity
  const uint32_t* nwork_dev;  // non-null: the count is read from device memory (leftovers of a previous stage)
  const uint32_t* wbeg_dev;   // non-null: the first list position of this launch, read from device memory — the leftovers ONE launch of the
                              // stage in front appended, [*wbeg_dev, *nwork_dev), aligned beside that stage's next launch (round 6)
  uint32_t nwork;
  int32_t* score;
  int32_t* status;
  uint8_t* cigar_ops;
  const int64_t* cigar_off;
  int64_t* cigar_begin;
  int32_t* cigar_len;
  int32_t* hist;          // FULL: history, one slice per workgroup
  int64_t hist_stride;    // ints per workgroup
  uint32_t* fb_list;
  uint32_t* fb_count;
  int g;                  // score step = gcd(x, o+e, e)
  int x, oe, e;           // penalties (score units), for the backtrace
  int oe2, e2;            // gap-affine-2p: o2 + e2 and e2 (0 = gap-affine)
  int min_wf_len, max_dist_thr, steps_between;
  int heur;               // ADAPT instantiations: 1 = wf-adaptive, 2 = X-drop (R/wavefront_heuristic.c:297-383)
  int xdrop;
  int max_steps;          // INT_MAX = unlimited (R/wavefront_unialign.c:98-107)
  int scope;              // max_score_scope (R/wavefront_components.c:81-124): null steps beyond it end the alignment "unreachable"
  int lds_words;          // SEQLDS: words reserved per sequence in dynamic LDS
  int split;              // FULL: 1 = history slot per PAIR of this launch + end states; the backtrace runs in its own
                          //       thread-per-alignment kernel afterwards (latency of 64 walks overlapped per wave)
  uint32_t work_begin;    // split: first work item of this launch (slot = item - work_begin)
  int4* end_state;        // split: per slot {end score, end k, end offset, 1 = walk it}
  int ef, pbf, pef, tbf, tef;  // ends-free span with these free ends (R/wavefront_termination.c:115-162)
  int h16;                // FULL: 1 = history entries are 4 x int16 (sequences < 32000 bases) instead of 4 x int32
  int32_t* done;          // single-call path: per pair, set to 1 (system scope) after everything else of the pair was written:
                          // the host polls it in the pinned block instead of waiting for the stream
  int debug;              // timing experiments only: 1 = skip the backtrace, 2 = skip the history stores
  long long pb_code_ints; // piggy-back: ints of a slot reserved for the code records; then pb_event_ints of event bytes, then the runs
  long long pb_event_ints;
  int pb;                 // FULL + split: 1 = piggy-back history: one byte of origin codes per (step, diagonal)
                          // instead of the offsets; the walk follows the codes and the matches are re-extended afterwards
  const uint2* lane_codes; // wfa_lane_kernel<.., FULL>: the comparison bits of every wave-step (64 lanes x 8 bytes per record)
  int seg_w;              // > 0: the history was written by wfa_seg_kernel<.., FULL>: piggy-back code records of seg_w bytes, the byte of
                          // diagonal k at k mod seg_w (a.pb = 1)
  int pb_raw;             // piggy-back codes written by wfa_slim_kernel: the four comparison bits as they fall out of the subtractions
                          // (bit 3: mismatch below the best gap, 2: deletion below insertion, 1 / 0: extension of I / D below its
                          // opening); the walk maps them to the codes above through a 16-entry table
  int win;                // 1: the sequences do not fit LDS — wfa_slim_kernel's windowed form stages lds_words words of each and moves the windows along
  const uint32_t* one;    // host side only: the single pair's SlimOne block (wfa_slim.hpp) — launch_slim_shape passes it as a kernel argument
  uint32_t* dbg;          // counting builds (-DWFA_SLIM_COUNTERS=1) with WFA_HIP_STAGE_TIMING=1: eight counters of wfa_slim_kernel
  int slim;               // 1: launches that fit wfa_slim_kernel (wfa_slim.hpp: 128 diagonals, gap-affine, wf-adaptive, end-to-end,
                          // sequences in LDS, score-only or piggy-back split history) take it instead of wfa_band_kernel (same results)
};

// One pair per call (wfa_slim.hpp: wfa_slim_kernel_one / wfa_slim_kernel_mailbox): everything the wave reads from the host about the pair
#define WFA_SLIM_ONE_WORDS 136   // 2 x (1000 bases + the look-ahead words)
struct SlimOne { uint32_t w[8 + WFA_SLIM_ONE_WORDS]; };   // [0,4) WfaPairMeta, [4,8) cigar_off[0..1], [8,..) words
// The mailbox of the resident one-pair kernel (pinned host memory; host and device sides on lines of their own).  The request travels
// as WFA_MB_LINES lines of 64 B: 15 words of the SlimOne block + the request number, so that the line is its own "ready" flag
#define WFA_MB_LINES 10   // 150 >= 8 + WFA_SLIM_ONE_WORDS words
struct SlimMailbox {
  uint32_t req[WFA_MB_LINES][16];   // host -> device
  uint32_t quit;         // host -> device: leave now
  uint32_t idle_ticks;   // leave after this many ticks of the 100 MHz clock without a request
  uint32_t pad0[14];
  unsigned long long done;   // device -> host, ONE 8-byte store: bits 0-23 the last request served, 24-31 its status code (0 done, 2 step limit,
                             // 255 handed on), 32-63 its score
  uint32_t served;       // requests served by this instance (diagnostics)
  uint32_t ticks;        // 10 ns ticks the last request took on the device, arrival to answer (diagnostics)
  uint32_t pad1[12];
  uint32_t alive;        // 1 from the host's launch of an instance until the instance leaves (the instance's last store)
  uint32_t pad2[15];
};


// index of the lowest set bit, ~0u for 0 (v_ffbl_b32 semantics)
__device__ __forceinline__ uint32_t band_ffbl(uint32_t x) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// wave-wide minimum without LDS traffic: butterfly inside each row of 16 lanes with DPP, then 4 row leaders
__device__ __forceinline__ int wave_min_dpp(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ int wave_max_dpp(int v) { return -wave_min_dpp(-v); }

template <int NCH>
struct Band {
  static constexpr int W = 64 * NCH;
  static constexpr int WI = (NCH == 3) ? 256 : W;  // records are indexed by k mod WI (a power of two)
  static constexpr int REC = 4 * WI;  // history record of one score: {M, I, D, window base} per window position (16 B)

  // value of the diagonal below / above across chunk boundaries
  static __device__ __forceinline__ int below(const int (&r)[NCH], int c, int nullv = WFA_OFFSET_NULL) {
    int fill = nullv;
    if (c > 0) fill = __builtin_amdgcn_readlane(r[c - 1], 63);
    return __builtin_amdgcn_update_dpp(fill, r[c], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  }
  static __device__ __forceinline__ int above(const int (&r)[NCH], int c, int nullv = WFA_OFFSET_NULL) {
    int fill = nullv;
    if (c < NCH - 1) fill = __builtin_amdgcn_readlane(r[c + 1], 0);
    return __builtin_amdgcn_update_dpp(fill, r[c], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  }
  // positions (0..W-1) of the first / last set bit over the per-chunk ballots; -1 / W if none
  static __device__ __forceinline__ int first_pos(const unsigned long long (&b)[NCH]) {
    int p = W;
#pragma unroll
    for (int c = NCH - 1; c >= 0; --c) if (b[c]) p = c * 64 + (int)__builtin_ctzll(b[c]);
    return p;
  }
  static __device__ __forceinline__ int last_pos(const unsigned long long (&b)[NCH]) {
    int p = -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) if (b[c]) p = c * 64 + 63 - (int)__builtin_clzll(b[c]);
    return p;
  }
  // shift a register set by `delta` window positions: new[pos] = old[pos + delta] (NULL outside)
  static __device__ __forceinline__ void shift(int (&r)[NCH], int delta, int lane, int nullv = WFA_OFFSET_NULL) {
    int out[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = c * 64 + lane + delta;
      int v = nullv;
#pragma unroll
      for (int sc = 0; sc < NCH; ++sc) {
        const int t = __shfl(r[sc], q & 63, 64);
        if ((q >> 6) == sc) v = t;
      }
      out[c] = v;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) r[c] = out[c];
  }
};

// One 16-byte load fetches {M, I, D, window base} of (score index, diagonal): the address depends on
// (si, k) alone, so all candidates of a backtrace step are in flight together (one memory round trip).
// (16-bit form: every negative offset is stored as -1, every offset and diagonal fits in 15 bits.)
__device__ __forceinline__ int sat16(int v) { return ((unsigned)v > 32767u) ? -1 : v; }  // inputs of a valid cell never exceed it
template <int NCH>
__device__ __forceinline__ int4 band_entry(const int* hist, int si, int k, int h16) {
  typedef Band<NCH> BD;
  int4 e = make_int4(WFA_OFFSET_NULL, WFA_OFFSET_NULL, WFA_OFFSET_NULL, 0);
  if (si >= 0) {
    if (h16) {
      const short4 q = reinterpret_cast<const short4*>(hist + (long long)si * (BD::REC / 2))[k & (BD::WI - 1)];
      e = make_int4(q.x < 0 ? WFA_OFFSET_NULL : q.x, q.y < 0 ? WFA_OFFSET_NULL : q.y, q.z < 0 ? WFA_OFFSET_NULL : q.z, q.w);
    } else {
      e = reinterpret_cast<const int4*>(hist + (long long)si * BD::REC)[k & (BD::WI - 1)];
    }
    if (k < e.w || k >= e.w + BD::W) { e.x = WFA_OFFSET_NULL; e.y = WFA_OFFSET_NULL; e.z = WFA_OFFSET_NULL; }
  }
  return e;
}
__device__ __forceinline__ long long band_pack(int si, int o, int add, int type) {
  // a candidate at a negative score does not exist (R/wavefront_backtrace.c:64-219 return NULL)
  return (si < 0) ? (long long)WFA_OFFSET_NULL : ((((long long)(o + add)) << 4) | type);
}

// R/wavefront_backtrace.c:320-529 for gap-affine over the band history; single lane.  Scores are walked
// as step indices si = s / g (x, o+e, e are multiples of g).
template <int NCH>
__device__ void band_backtrace(const int* hist, const BandArgs& a, int plen, int tlen, int end_s, int end_k,
                               int end_off, uint8_t* buf, long long* begin_out, int lane, int nlanes,
                               uint32_t* runs_top = nullptr, int* nruns_out = nullptr) {
  // nlanes = 64: executed by the whole wave with uniform control flow (every lane walks the same path, the
  // loads are broadcasts, runs of one op are stored 64 bytes at a time); nlanes = 1: one thread per alignment
  // runs_top != nullptr: instead of op bytes, emit (length << 8 | op) run records downwards from runs_top
  // (the top of this pair's history slot: records above the walk's current score are dead) — the bytes
  // are written afterwards, coalesced, by wfa_band_expand_kernel
  typedef Band<NCH> BD;
  struct Ent { int m, i1, d1, i2, d2; };
  int nruns = 0;
  const bool two = a.oe2 > 0;
  // entry of (score index, diagonal): the banded kernel's window records (gap-affine: {M, I, D, base}; 2p: 16 bytes of
  // int16 halves {M | I1, D1 | base, I2 | D2, -})
  auto entry = [&](int si, int kk) -> Ent {
    Ent e = {WFA_OFFSET_NULL, WFA_OFFSET_NULL, WFA_OFFSET_NULL, WFA_OFFSET_NULL, WFA_OFFSET_NULL};
    if (si < 0) return e;
    auto nz = [](int v) { return v < 0 ? WFA_OFFSET_NULL : v; };
    if (two) {
      const int4 q = reinterpret_cast<const int4*>(hist + (long long)si * BD::REC)[kk & (BD::WI - 1)];
      const int base = q.y >> 16;
      if (kk >= base && kk < base + BD::W) {
        e.m = nz((int)(short)(q.x & 0xffff)); e.i1 = nz(q.x >> 16); e.d1 = nz((int)(short)(q.y & 0xffff));
        e.i2 = nz((int)(short)(q.z & 0xffff)); e.d2 = nz(q.z >> 16);
      }
    } else {
      const int4 q = band_entry<NCH>(hist, si, kk, a.h16);
      e.m = q.x; e.i1 = q.y; e.d1 = q.z;
    }
    return e;
  };
  auto push = [&](long long& bg, char c, int n) {
    if (runs_top) { if (n > 0) { *(runs_top - nruns) = ((uint32_t)n << 8) | (uint8_t)c; ++nruns; } }
    else for (int i = lane; i < n; i += nlanes) buf[bg - 1 - i] = (uint8_t)c;
    bg -= n;
  };
  enum { BT_I1_OPEN = 1, BT_I1_EXT = 2, BT_I2_OPEN = 3, BT_I2_EXT = 4, BT_D1_OPEN = 5, BT_D1_EXT = 6, BT_D2_OPEN = 7, BT_D2_EXT = 8, BT_M = 9 };
  const int dx = a.x / a.g, doe = a.oe / a.g, de = a.e / a.g;
  const int doe2 = two ? a.oe2 / a.g : 0, de2 = two ? a.e2 / a.g : 0;
  long long begin = (long long)plen + tlen;
  int comp = 0, si = end_s / a.g, k = end_k, offset = end_off;
  int h = offset, v = offset - k;
  push(begin, 'D', max(plen - v, 0));
  push(begin, 'I', max(tlen - h, 0));
  while (v > 0 && h > 0 && si > 0) {
    const int si_x = si - dx, si_o = si - doe, si_e = si - de;
    const int si_o2 = two ? si - doe2 : -1, si_e2 = two ? si - de2 : -1;
    long long best;
    if (comp == 0) {
      const Ent ex = entry(si_x, k);
      const Ent eol = entry(si_o, k - 1), eoh = entry(si_o, k + 1);
      const Ent eel = entry(si_e, k - 1), eeh = entry(si_e, k + 1);
      const long long c0 = band_pack(si_x, ex.m, 1, BT_M);
      const long long c1 = band_pack(si_o, eol.m, 1, BT_I1_OPEN), c2 = band_pack(si_e, eel.i1, 1, BT_I1_EXT);
      const long long c3 = band_pack(si_o, eoh.m, 0, BT_D1_OPEN), c4 = band_pack(si_e, eeh.d1, 0, BT_D1_EXT);
      best = max(max(c0, max(c1, c2)), max(c3, c4));
      if (two) {
        const Ent fol = entry(si_o2, k - 1), foh = entry(si_o2, k + 1);
        const Ent fel = (de2 == de) ? eel : entry(si_e2, k - 1), feh = (de2 == de) ? eeh : entry(si_e2, k + 1);
        best = max(best, max(band_pack(si_o2, fol.m, 1, BT_I2_OPEN), band_pack(si_e2, fel.i2, 1, BT_I2_EXT)));
        best = max(best, max(band_pack(si_o2, foh.m, 0, BT_D2_OPEN), band_pack(si_e2, feh.d2, 0, BT_D2_EXT)));
      }
    } else if (comp == 1) {
      best = max(band_pack(si_o, entry(si_o, k - 1).m, 1, BT_I1_OPEN), band_pack(si_e, entry(si_e, k - 1).i1, 1, BT_I1_EXT));
    } else if (comp == 2) {
      best = max(band_pack(si_o, entry(si_o, k + 1).m, 0, BT_D1_OPEN), band_pack(si_e, entry(si_e, k + 1).d1, 0, BT_D1_EXT));
    } else if (comp == 3) {
      best = max(band_pack(si_o2, entry(si_o2, k - 1).m, 1, BT_I2_OPEN), band_pack(si_e2, entry(si_e2, k - 1).i2, 1, BT_I2_EXT));
    } else {
      best = max(band_pack(si_o2, entry(si_o2, k + 1).m, 0, BT_D2_OPEN), band_pack(si_e2, entry(si_e2, k + 1).d2, 0, BT_D2_EXT));
    }
    if (best < 0) break;
    if (comp == 0) {
      const int src = (int)(best >> 4);
      push(begin, 'M', offset - src);
      offset = src;
      v = offset - k; h = offset;
      if (v <= 0 || h <= 0) break;
    }
    const int type = (int)(best & 0xF);
    if (type == BT_M) { si = si_x; comp = 0; push(begin, 'X', 1); --offset; }
    else if (type == BT_I1_OPEN) { si = si_o; comp = 0; push(begin, 'I', 1); --k; --offset; }
    else if (type == BT_I1_EXT) { si = si_e; comp = 1; push(begin, 'I', 1); --k; --offset; }
    else if (type == BT_I2_OPEN) { si = si_o2; comp = 0; push(begin, 'I', 1); --k; --offset; }
    else if (type == BT_I2_EXT) { si = si_e2; comp = 3; push(begin, 'I', 1); --k; --offset; }
    else if (type == BT_D1_OPEN) { si = si_o; comp = 0; push(begin, 'D', 1); ++k; }
    else if (type == BT_D1_EXT) { si = si_e; comp = 2; push(begin, 'D', 1); ++k; }
    else if (type == BT_D2_OPEN) { si = si_o2; comp = 0; push(begin, 'D', 1); ++k; }
    else { si = si_e2; comp = 4; push(begin, 'D', 1); ++k; }
    v = offset - k; h = offset;
  }
  if (comp == 0) {
    if (v > 0 && h > 0) {
      const int n = min(v, h);
      push(begin, 'M', n);
      v -= n; h -= n;
    }
    for (; v > 0; --v) push(begin, 'D', 1);
    for (; h > 0; --h) push(begin, 'I', 1);
  }
  *begin_out = begin;
  if (nruns_out) *nruns_out = nruns;
}

template <int NCH, bool FULL, bool ADAPT, bool SEQLDS, bool PB, bool SPLIT, int X, int OE, int E, int OE2, int E2>
__device__ __forceinline__ void wfa_band_body(const BandArgs& a) {
  static_assert(FULL || !SPLIT, "split launches have a history");
  static_assert(SPLIT || !PB, "piggy-back history: split launches only");
  static_assert(FULL || !PB, "piggy-back history only with a history");
  constexpr bool TWO = OE2 > 0;  // gap-affine-2p: second pair of gap components (R/wavefront_compute_affine2p.c:45-106)
  constexpr int E2D = TWO ? E2 : 1;
  typedef Band<NCH> BD;
  constexpr int W = BD::W;
  constexpr int WI = BD::WI;
  constexpr int DM1 = (X > OE) ? X : OE;
  // M history: depths 1..DM in registers; 2p: the depths beyond max(X, OE), read only once (at OE2), are kept as int16
  // pairs, two depths per register, shifted with one v_alignbit each (half the registers and moves; reads < 32000 bases)
  constexpr int DM = TWO ? DM1 : ((DM1 > OE2) ? DM1 : OE2);  // (X, OE, E, OE2, E2) = (x, o1 + e1, e1, o2 + e2, e2) / g
  constexpr int NP = TWO ? (OE2 - DM + 1) / 2 + 1 : 1;       // packed registers: depths DM+1 .. OE2 (+ slack)
  static_assert(!TWO || OE2 > DM1, "2p: o2 + e2 is the deepest history read");
  extern __shared__ uint32_t slds[];
  uint32_t* const sP = slds;
  uint32_t* const sT = slds + a.lds_words;
  const int lane = threadIdx.x;
  int* hist = FULL ? a.hist + (long long)blockIdx.x * a.hist_stride : nullptr;
  const int rec_ints = PB ? WI / 4 : ((a.h16 && !TWO) ? BD::REC / 2 : BD::REC);  // 2p: 16-byte entries of 6 x int16  // piggy-back: one byte per window position
  const int max_records = FULL ? (int)min((long long)INT_MAX, (PB ? a.pb_code_ints : a.hist_stride) / rec_ints) : INT_MAX;

  constexpr bool split = SPLIT;  // (a template parameter: the in-kernel walk of the other form costs 20 VGPRs = 2 waves per SIMD)
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  const uint32_t w0 = split ? a.work_begin : 0u;
  const uint32_t wb = a.wbeg_dev ? *a.wbeg_dev : 0u;
  for (uint32_t wi = w0 + wb + blockIdx.x; wi < w0 + nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    const WfaPairMeta pm = a.meta[pair];
    const int plen = pm.plen, tlen = pm.tlen;
    const int ak = tlen - plen;
    if (split) hist = a.hist + (long long)(wi - w0) * a.hist_stride;
    const uint32_t* gP = a.words + pm.p_woff;
    const uint32_t* gT = a.words + pm.t_woff;
    const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
    bool fallback = false;
    if (SEQLDS) {
      if (nwp + 3 > a.lds_words || nwt + 3 > a.lds_words) fallback = true;
      else {
        __syncthreads();
        for (int i = lane; i < nwp + 3; i += 64) sP[i] = (i < nwp) ? gP[i] : 0u;
        for (int i = lane; i < nwt + 3; i += 64) sT[i] = (i < nwt) ? gT[i] : 0u;
        __syncthreads();
      }
    }
    int B = -(W / 2);  // diagonal of window position 0; the window then follows the live diagonals
    if (a.ef) {
      // wavefront 0 spans the diagonals [-pattern_begin_free, text_begin_free] (R/wavefront_aligner.c:259-302)
      if (a.pbf + a.tbf + 1 > W - 20) fallback = true;
      B = (a.tbf - a.pbf) / 2 - W / 2;
    }
    int result = 0;
    int end_k = 0, end_off = 0, end_s = 0;
    int stop_status = 0, stop_score = 0;  // ended without reaching the end cell: dropped (status 1) or step limit (-100)
    if (!fallback) {
      // per lane: lim = min(tlen, plen + k) (in-bounds <=> offset <= lim; lim - offset = longest possible run),
      // dlim = max(tlen, plen + k) (dlim - offset = distance to the end, R/wavefront_heuristic.c:176-192)
      int kk[NCH], lim[NCH], dlim[NCH], cur[NCH], Mh[DM][NCH], Ih[E][NCH], Dh[E][NCH], I2h[E2D][NCH], D2h[E2D][NCH], PH[NP][NCH];
      int code[PB ? NCH : 1];  // piggy-back: origin of M (bits 0-1: 0 mismatch, 1 deletion, 2 insertion), of I (bit 2: extension) and of D (bit 3);
                               // 2p: bits 0-2 origin of M (0 mismatch, 1 D1, 2 D2, 3 I1, 4 I2), bit 3 / 4 / 5 / 6: I1 / D1 / I2 / D2 came by extension
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (PB) code[c] = 0;
        kk[c] = B + c * 64 + lane;
        lim[c] = min(tlen, plen + kk[c]); dlim[c] = max(tlen, plen + kk[c]);
        cur[c] = (kk[c] == 0) ? 0 : WFA_OFFSET_NULL;  // wavefront 0
        if (a.ef && kk[c] >= -a.pbf && kk[c] <= a.tbf) cur[c] = max(kk[c], 0);
#pragma unroll
        for (int j = 0; j < E; ++j) { Ih[j][c] = WFA_OFFSET_NULL; Dh[j][c] = WFA_OFFSET_NULL; }
#pragma unroll
        for (int j = 0; j < E2D; ++j) { I2h[j][c] = WFA_OFFSET_NULL; D2h[j][c] = WFA_OFFSET_NULL; }
#pragma unroll
        for (int j = 0; j < DM; ++j) Mh[j][c] = WFA_OFFSET_NULL;
#pragma unroll
        for (int j = 0; j < NP; ++j) PH[j][c] = -1;  // both halves NULL
      }
      int s = 0, steps_wait = a.steps_between, dead_steps = 0;
      int have_max_sw = 0, max_sw = 0;      // X-drop state (R/wavefront_heuristic.c:114-121)
      int last_nonnull = 0;                 // last score whose compute-next had a non-null input (score 0 counts)
      bool done = false;
      // Chunks that can hold a live diagonal: when the hull of all ring registers fits the first ACT_SMALL chunks (with the
      // margins of the window check) the other chunks' registers are all NULL and stay so, and the two heavy blocks of a
      // step, extension and compute-next, run on the active chunks only (one straight-line body per form, chosen by a
      // wave-uniform branch; the rest of the step is common).  10 kb at 8 % with wf-adaptive: the wavefront is ~34
      // diagonals wide on average, so most steps of the 128-diagonal form run on 64 lanes.  Decided at the hull check
      // (every 8 steps).
      constexpr int ACT_SMALL = (!TWO && NCH == 2) ? 1 : ((!TWO && NCH == 4) ? 2 : NCH);
      bool big = true;
      for (int step = 0;; ++step) {
        if (PB) {
          // piggy-back history of score s: the origin codes do not depend on the extension or the cut-off, and the walk
          // starts from the codes of the END cell, so they are stored before the termination test
          const int si = step;  // = s / g: the score grows by g per step
          if (si + 1 >= max_records) { fallback = true; break; }
          uint8_t* rec = reinterpret_cast<uint8_t*>(hist) + (long long)si * WI;
#pragma unroll
          for (int c = 0; c < NCH; ++c) rec[kk[c] & (WI - 1)] = (uint8_t)code[c];
        }
        // ---------------- extend M[s] ----------------
        unsigned long long live[NCH];
        bool any_live = false;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { live[c] = __ballot(cur[c] >= 0); any_live |= (live[c] != 0); }
        if (any_live) {
          dead_steps = 0;
          {
            auto extend_chunks = [&](auto act_tag) {
              constexpr int ACT = decltype(act_tag)::value;
              // all chunks advance together: 32 bases per iteration (three packed words per sequence)
              int h[NCH], v[NCH], left[NCH];
              bool any_more = false;
  #pragma unroll
              for (int c = 0; c < ACT; ++c) {
                h[c] = max(cur[c], 0); v[c] = max(cur[c] - kk[c], 0);
                left[c] = (cur[c] >= 0) ? lim[c] - cur[c] : 0;
                any_more |= left[c] > 0;
              }
              if (__any(any_more)) {
                bool more;
                do {
                  more = false;
  #pragma unroll
                  for (int c = 0; c < ACT; ++c) {
                    const int pi = v[c] >> 4, ti = h[c] >> 4;
                    uint32_t p0, p1, p2, t0, t1, t2;
                    if (SEQLDS) { p0 = sP[pi]; p1 = sP[pi + 1]; p2 = sP[pi + 2]; t0 = sT[ti]; t1 = sT[ti + 1]; t2 = sT[ti + 2]; }
                    else { p0 = gP[pi]; p1 = gP[pi + 1]; p2 = gP[pi + 2]; t0 = gT[ti]; t1 = gT[ti + 1]; t2 = gT[ti + 2]; }
                    const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v[c] << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h[c] << 1);
                    const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v[c] << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h[c] << 1);
                    // first differing bit of xh:xl (v_ffbl_b32 gives ~0 for 0: `| 32` is +32 or stays ~0)
                    const uint32_t fb = min(band_ffbl(xl), band_ffbl(xh) | 32u);
                    const int m = min((int)(fb >> 1), min(32, left[c]));
                    v[c] += m; h[c] += m; left[c] -= m;
                    more |= (m == 32) && (left[c] > 0);
                  }
                } while (__any(more));
  #pragma unroll
                for (int c = 0; c < ACT; ++c) if (cur[c] >= 0) cur[c] = h[c];
              }
            };
            if (ACT_SMALL < NCH && !big) extend_chunks(band_int<ACT_SMALL>{});
            else extend_chunks(band_int<NCH>{});
          }
          // ---------------- termination ----------------
          if (a.ef) {
            // ends-free: the lowest diagonal that touches a border within the free budget wins
            unsigned long long hit[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const int h = cur[c], v = cur[c] - kk[c];
              hit[c] = __ballot(cur[c] >= 0 && ((h >= tlen && plen - v <= a.pef) || (v >= plen && tlen - h <= a.tef)));
            }
            const int hp = BD::first_pos(hit);
            if (hp < W) {
              int eo = 0;
#pragma unroll
              for (int c = 0; c < NCH; ++c) if ((hp >> 6) == c) eo = __builtin_amdgcn_readlane(cur[c], hp & 63);
              done = true; result = -s; end_k = B + hp; end_off = eo; end_s = s; break;
            }
          } else {
            const int p = ak - B;
            int at_end = WFA_OFFSET_NULL;
#pragma unroll
            for (int c = 0; c < NCH; ++c) if ((p >> 6) == c) at_end = __builtin_amdgcn_readlane(cur[c], p & 63);
            if (p >= 0 && p < W && at_end >= tlen) { done = true; result = -s; end_k = ak; end_off = tlen; end_s = s; break; }
          }
          // ---------------- wf-adaptive cut-off (R/wavefront_heuristic.c:257-293,509-567) ----------------
          if (ADAPT && a.heur == 2) {
            // X-drop (R/wavefront_heuristic.c:297-383; match = 0: the "score" of a cell is (-(v + h) - s) / 2, C division)
            --steps_wait;
            if (steps_wait <= 0) {
              const int lo = B + BD::first_pos(live), hi = B + BD::last_pos(live);
              int sw[NCH], cmax = -0x40000000;   // (not INT_MIN: the wave maximum negates it)
#pragma unroll
              for (int c = 0; c < NCH; ++c) {
                sw[c] = (-(2 * cur[c] - kk[c]) - s) / 2;
                if (cur[c] >= 0) cmax = max(cmax, sw[c]);
              }
              cmax = wave_max_dpp(cmax);
              if (have_max_sw) {
                unsigned long long ok[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) ok[c] = __ballot(cur[c] >= 0 && max_sw - sw[c] < a.xdrop);
                const int fp = BD::first_pos(ok), lp = BD::last_pos(ok);
                const int new_lo = (fp < W) ? B + fp : hi + 1, new_hi = (fp < W) ? B + lp : hi;
                if (new_lo != lo || new_hi != hi) {
#pragma unroll
                  for (int c = 0; c < NCH; ++c) {
                    const bool drop = kk[c] < new_lo || kk[c] > new_hi;
                    if (drop) { cur[c] = WFA_OFFSET_NULL; Ih[0][c] = WFA_OFFSET_NULL; Dh[0][c] = WFA_OFFSET_NULL; I2h[0][c] = WFA_OFFSET_NULL; D2h[0][c] = WFA_OFFSET_NULL; }
                  }
                }
                if (cmax > max_sw) max_sw = cmax;
              } else {
                max_sw = cmax; have_max_sw = 1;
              }
              steps_wait = a.steps_between;
            }
          } else if (ADAPT) {
            --steps_wait;
            if (steps_wait <= 0) {
              const int lo = B + BD::first_pos(live), hi = B + BD::last_pos(live);
              if (hi - lo + 1 >= a.min_wf_len) {
                int d[NCH], dmin = max(plen, tlen);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                  d[c] = (cur[c] >= 0) ? dlim[c] - cur[c] : -WFA_OFFSET_NULL;  // max(plen - v, tlen - h)
                  dmin = min(dmin, d[c]);
                }
                dmin = wave_min_dpp(dmin);
                unsigned long long ok[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) ok[c] = __ballot(d[c] - dmin <= a.max_dist_thr);
                const int fp = BD::first_pos(ok), lp = BD::last_pos(ok);
                const int lc = (fp < W) ? B + fp : INT_MAX, hc = (lp >= 0) ? B + lp : INT_MIN;
                int new_lo = lo, new_hi = hi;
                const int top_limit = min(ak, hi);
                if (top_limit > lo) new_lo = min(lc, top_limit);
                const int bottom_limit = max(ak, new_lo);
                if (bottom_limit < hi) new_hi = max(hc, bottom_limit);
                steps_wait = a.steps_between;
                if (new_lo != lo || new_hi != hi) {
#pragma unroll
                  for (int c = 0; c < NCH; ++c) {
                    const bool drop = kk[c] < new_lo || kk[c] > new_hi;
                    if (drop) { cur[c] = WFA_OFFSET_NULL; Ih[0][c] = WFA_OFFSET_NULL; Dh[0][c] = WFA_OFFSET_NULL; I2h[0][c] = WFA_OFFSET_NULL; D2h[0][c] = WFA_OFFSET_NULL; }
                  }
                }
              }
            }
          }
        } else {
          // nothing alive at this score; if the whole ring is dead the reference ends "unreachable"
          // after its null-step count runs out: leave that rare case to the general kernel
          // (without a heuristic that cannot last: leave it to the general kernel; with one, the null-step count below ends it)
          if (!ADAPT && ++dead_steps > 2 * DM + 2 + (TWO ? OE2 : 0)) { fallback = true; break; }
        }
        // ---------------- history of score s (after the cut-off, so dropped lanes read NULL) ----------------
        const int si = step;  // = s / g
        if (FULL && !PB) {
          if (si + 1 >= max_records) { fallback = true; break; }
          int* rec = hist + (long long)si * rec_ints;
          if (!(a.debug & 2)) {
            if (TWO) {
              // {M | I1 << 16, D1 | B << 16, I2 | D2 << 16, 0}: int16 halves, negative -> -1 (reads < 32000 bases)
#pragma unroll
              for (int c = 0; c < NCH; ++c)
                reinterpret_cast<int4*>(rec)[kk[c] & (WI - 1)] =
                    make_int4((sat16(cur[c]) & 0xffff) | (sat16(Ih[0][c]) << 16), (sat16(Dh[0][c]) & 0xffff) | (B << 16),
                              (sat16(I2h[0][c]) & 0xffff) | (sat16(D2h[0][c]) << 16), 0);
            } else if (a.h16) {
#pragma unroll
              for (int c = 0; c < NCH; ++c)
                reinterpret_cast<short4*>(rec)[kk[c] & (WI - 1)] =
                    make_short4((short)sat16(cur[c]), (short)sat16(Ih[0][c]), (short)sat16(Dh[0][c]), (short)B);
            } else {
#pragma unroll
              for (int c = 0; c < NCH; ++c) reinterpret_cast<int4*>(rec)[kk[c] & (WI - 1)] = make_int4(cur[c], Ih[0][c], Dh[0][c], B);
            }
          }
        }
        // ---------------- keep the ring inside the window (every HP steps; growth is <= 1 diagonal/step) ----
        constexpr int HP = 8;   // period of the check (4 keeps the small form a little longer but measures slower on 10 kb reads)
        if ((step & (HP - 1)) == 0) {
          unsigned long long hull[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            int any = cur[c];
#pragma unroll
            for (int j = 0; j < E; ++j) any &= Ih[j][c] & Dh[j][c];
            if (TWO) {
#pragma unroll
              for (int j = 0; j < E2D; ++j) any &= I2h[j][c] & D2h[j][c];
            }
#pragma unroll
            for (int j = 0; j < DM - 1; ++j) any &= Mh[j][c];
            if (TWO) {
              any &= Mh[DM - 1][c];
#pragma unroll
              for (int j = 0; j < NP; ++j) any &= PH[j][c] & (PH[j][c] << 16);  // sign set iff both halves are NULL
            }
            hull[c] = __ballot(any >= 0);  // some register of this diagonal is not negative
          }
          const int fp = BD::first_pos(hull), lp = BD::last_pos(hull);
          if (lp >= 0) {
            const int width = lp - fp + 1;
            if (width > ((a.debug >> 8) ? (a.debug >> 8) : W - 20)) { fallback = true; break; }  // (debug >> 8: width experiments)
            // small form: the hull (and HP steps of growth either way) fits the first ACT_SMALL chunks, lane 63 of the last
            // active chunk included in the margin, so nothing can reach the other chunks before the next check; a little
            // hysteresis keeps a hull near the limit from being shifted to and fro
            constexpr int WS = 64 * ACT_SMALL;
            const bool want_small = ACT_SMALL < NCH && !(a.debug & 4) && width <= (big ? WS - 2 * HP - 6 : WS - 2 * HP - 2);
            const int hi_lim = (want_small ? WS : W) - HP - 2;
            if (fp < HP + 1 || lp > hi_lim) {
              // re-centre in the window (or in its small form)
              int delta = fp - ((want_small ? WS : W) - width) / 2;
              B += delta;
#pragma unroll
              for (int c = 0; c < NCH; ++c) { kk[c] += delta; lim[c] = min(tlen, plen + kk[c]); dlim[c] = max(tlen, plen + kk[c]); }
              BD::shift(cur, delta, lane);
#pragma unroll
              for (int j = 0; j < E; ++j) { BD::shift(Ih[j], delta, lane); BD::shift(Dh[j], delta, lane); }
              if (TWO) {
#pragma unroll
                for (int j = 0; j < E2D; ++j) { BD::shift(I2h[j], delta, lane); BD::shift(D2h[j], delta, lane); }
              }
#pragma unroll
              for (int j = 0; j < DM - 1; ++j) BD::shift(Mh[j], delta, lane);
              if (TWO) {
                BD::shift(Mh[DM - 1], delta, lane);
#pragma unroll
                for (int j = 0; j < NP; ++j) BD::shift(PH[j], delta, lane, -1);
              }
            }
            if (ACT_SMALL < NCH) {
              big = !want_small;
              if (want_small) {
                // the oldest M of the now inactive chunks is outside the hull test (compute-next drops it) but is read as a
                // neighbour by the last active chunk: NULL
#pragma unroll
                for (int c = ACT_SMALL; c < NCH; ++c) Mh[DM - 1][c] = WFA_OFFSET_NULL;
              }
            }
          }
        }
        // ---------------- compute-next for score s+g ----------------
        int insig = -1;  // AND of all inputs: non-negative iff some input offset is not NULL-ish
        auto compute_next = [&](auto act_tag) {
          constexpr int ACT = decltype(act_tag)::value;
          constexpr int ACTP = (ACT < NCH) ? ACT + 1 : NCH;   // (the neighbour of the last active chunk is read, as NULLs)
          if (TWO) {
            // the value leaving depth DM enters the packed ring: PH[0].lo = depth DM + 1, PH[0].hi = DM + 2, ...
#pragma unroll
            for (int j = NP - 1; j > 0; --j)
#pragma unroll
              for (int c = 0; c < ACT; ++c) PH[j][c] = (int)__builtin_amdgcn_alignbit((uint32_t)PH[j][c], (uint32_t)PH[j - 1][c], 16);
#pragma unroll
            for (int c = 0; c < ACT; ++c) PH[0][c] = (PH[0][c] << 16) | (max(Mh[DM - 1][c], -1) & 0xffff);
          }
#pragma unroll
          for (int j = DM - 1; j > 0; --j)
#pragma unroll
            for (int c = 0; c < ACT; ++c) Mh[j][c] = Mh[j - 1][c];
#pragma unroll
          for (int c = 0; c < ACT; ++c) Mh[0][c] = cur[c];
          s += a.g;
          insig = -1;  // AND of all inputs: non-negative iff some input offset is not NULL-ish
#pragma unroll
          for (int c = 0; c < ACT; ++c) {
            insig &= Mh[X - 1][c] & Mh[OE - 1][c] & Ih[E - 1][c] & Dh[E - 1][c];
            if (TWO) {
              constexpr int PD = TWO ? OE2 - 1 - DM : 0;  // depth OE2 (index OE2 - 1) sits in half PD & 1 of PH[PD / 2]
              const int mo2 = (PD & 1) ? (PH[PD / 2][c] >> 16) : (int)(short)(PH[PD / 2][c] & 0xffff);
              insig &= ((mo2 < 0) ? WFA_OFFSET_NULL : mo2) & I2h[E2D - 1][c] & D2h[E2D - 1][c];
            }
          }
          int ni[NCH], nd[NCH], nm[NCH], ni2[NCH], nd2[NCH];
          if (__any(insig >= 0)) {
            unsigned long long oob = 0;
            int gi[NCH], gd[NCH];
#pragma unroll
            for (int c = 0; c < ACTP; ++c) { gi[c] = max(Mh[OE - 1][c], Ih[E - 1][c]); gd[c] = max(Mh[OE - 1][c], Dh[E - 1][c]); }
#pragma unroll
            for (int c = 0; c < ACT; ++c) {
              int mo_lo = 0, ie_lo = 0, mo_hi = 0, de_hi = 0;
              if (PB) {
                mo_lo = BD::below(Mh[OE - 1], c); ie_lo = BD::below(Ih[E - 1], c);
                mo_hi = BD::above(Mh[OE - 1], c); de_hi = BD::above(Dh[E - 1], c);
                ni[c] = max(mo_lo, ie_lo) + 1;
                nd[c] = max(mo_hi, de_hi);
              } else {
                // I(k) = max(M_oe, I_e)(k-1) + 1, D(k) = max(M_oe, D_e)(k+1): the max commutes with the lane shift
                ni[c] = BD::below(gi, c) + 1;
                nd[c] = BD::above(gd, c);
              }
              ni2[c] = WFA_OFFSET_NULL; nd2[c] = WFA_OFFSET_NULL;
              int mo2_lo = 0, i2e_lo = 0, mo2_hi = 0, d2e_hi = 0;
              if (TWO) {
                constexpr int PD = TWO ? OE2 - 1 - DM : 0;
                const int plo = BD::below(PH[PD / 2], c, -1), phi = BD::above(PH[PD / 2], c, -1);
                const int m2lo = (PD & 1) ? (plo >> 16) : (int)(short)(plo & 0xffff), m2hi = (PD & 1) ? (phi >> 16) : (int)(short)(phi & 0xffff);
                mo2_lo = (m2lo < 0) ? WFA_OFFSET_NULL : m2lo; mo2_hi = (m2hi < 0) ? WFA_OFFSET_NULL : m2hi;
                i2e_lo = BD::below(I2h[E2D - 1], c); d2e_hi = BD::above(D2h[E2D - 1], c);
                ni2[c] = max(mo2_lo, i2e_lo) + 1;
                nd2[c] = max(mo2_hi, d2e_hi);
              }
              int m = max(max(nd[c], nd2[c]), max(Mh[X - 1][c] + 1, max(ni[c], ni2[c])));
              if (m > lim[c]) m = WFA_OFFSET_NULL;  // only M is clamped; negative values are dead already
              nm[c] = m;
              if (PB) {
                // the choice the backtrace would make (R/wavefront_backtrace.c:49-59: mismatch > D2 ext > D2 open > D1 ext >
                // D1 open > I2 ext > I2 open > I1 ext > I1 open on equal offsets), taken here where the candidates are in registers
                const int x1 = Mh[X - 1][c] + 1;
                if (TWO) {
                  const int best = max(max(nd[c], nd2[c]), max(x1, max(ni[c], ni2[c])));
                  const int mc = (x1 >= best) ? 0 : (nd2[c] >= best) ? 2 : (nd[c] >= best) ? 1 : (ni2[c] >= best) ? 4 : 3;
                  code[c] = mc | ((ie_lo >= mo_lo) ? 8 : 0) | ((de_hi >= mo_hi) ? 16 : 0) | ((i2e_lo >= mo2_lo) ? 32 : 0) | ((d2e_hi >= mo2_hi) ? 64 : 0);
                } else {
                  const int mc = (x1 >= max(nd[c], ni[c])) ? 0 : ((nd[c] >= ni[c]) ? 1 : 2);
                  code[c] = mc | ((ie_lo >= mo_lo) ? 4 : 0) | ((de_hi >= mo_hi) ? 8 : 0);
                }
              }
              oob |= __ballot(max(max(ni[c], nd[c]), max(ni2[c], nd2[c])) > lim[c]);
            }
            if (oob) {
              // trim the ends of I and D (R/wavefront_compute.c:571-605): outside [first,last] in-bounds -> NULL
              unsigned long long bi[NCH] = {}, bd[NCH] = {};   // (zero beyond the active chunks)
#pragma unroll
              for (int c = 0; c < ACT; ++c) {
                bi[c] = __ballot(ni[c] >= 0 && ni[c] <= lim[c]);
                bd[c] = __ballot(nd[c] >= 0 && nd[c] <= lim[c]);
              }
              const int ilo = BD::first_pos(bi), ihi = BD::last_pos(bi), dlo = BD::first_pos(bd), dhi = BD::last_pos(bd);
#pragma unroll
              for (int c = 0; c < ACT; ++c) {
                const int pos = c * 64 + lane;
                if (pos < ilo || pos > ihi) ni[c] = WFA_OFFSET_NULL;
                if (pos < dlo || pos > dhi) nd[c] = WFA_OFFSET_NULL;
              }
              if (TWO) {
#pragma unroll
                for (int c = 0; c < ACT; ++c) {
                  bi[c] = __ballot(ni2[c] >= 0 && ni2[c] <= lim[c]);
                  bd[c] = __ballot(nd2[c] >= 0 && nd2[c] <= lim[c]);
                }
                const int i2lo = BD::first_pos(bi), i2hi = BD::last_pos(bi), d2lo = BD::first_pos(bd), d2hi = BD::last_pos(bd);
#pragma unroll
                for (int c = 0; c < ACT; ++c) {
                  const int pos = c * 64 + lane;
                  if (pos < i2lo || pos > i2hi) ni2[c] = WFA_OFFSET_NULL;
                  if (pos < d2lo || pos > d2hi) nd2[c] = WFA_OFFSET_NULL;
                }
              }
            }
          } else {
#pragma unroll
            for (int c = 0; c < ACT; ++c) { ni[c] = WFA_OFFSET_NULL; nd[c] = WFA_OFFSET_NULL; nm[c] = WFA_OFFSET_NULL; ni2[c] = WFA_OFFSET_NULL; nd2[c] = WFA_OFFSET_NULL; }
          }
#pragma unroll
          for (int j = E - 1; j > 0; --j)
#pragma unroll
            for (int c = 0; c < ACT; ++c) { Ih[j][c] = Ih[j - 1][c]; Dh[j][c] = Dh[j - 1][c]; }
#pragma unroll
          for (int c = 0; c < ACT; ++c) { Ih[0][c] = ni[c]; Dh[0][c] = nd[c]; cur[c] = nm[c]; }
          if (TWO) {
#pragma unroll
            for (int j = E2D - 1; j > 0; --j)
#pragma unroll
              for (int c = 0; c < ACT; ++c) { I2h[j][c] = I2h[j - 1][c]; D2h[j][c] = D2h[j - 1][c]; }
#pragma unroll
            for (int c = 0; c < ACT; ++c) { I2h[0][c] = ni2[c]; D2h[0][c] = nd2[c]; }
          }
        };
        if (ACT_SMALL < NCH && !big) compute_next(band_int<ACT_SMALL>{});
        else compute_next(band_int<NCH>{});
        // ---------------- limits (R/wavefront_unialign.c:98-107, R/wavefront_extend.c:97-104) ----------------
        // The reference walks every integer score; here only multiples of g exist, the scores in between are null steps.
        // A run of more than `scope` null scores after the last non-null one ends the alignment "unreachable" at score
        // t = last + scope + 1; the step limit ends it at the first score >= max_steps; whichever comes first (the
        // limit is tested after compute-next of a score, the null-step count at its extension: the limit wins a tie).
        {
          const bool null_step = !__any(insig >= 0);
          const int t_unreach = last_nonnull + a.scope + 1;
          const bool unreach = ADAPT && (t_unreach < s || (t_unreach == s && null_step));
          const bool limit = s >= a.max_steps;
          if (limit && (!unreach || a.max_steps <= t_unreach)) { stop_status = WFA_STATUS_MAX_STEPS_REACHED; stop_score = -a.max_steps; break; }
          if (unreach) { stop_status = WFA_STATUS_PARTIAL; stop_score = FULL ? INT_MIN : -t_unreach; break; }
          if (!null_step) last_nonnull = s;
        }
        if (step > (1 << 24)) { fallback = true; break; }
      }
      if (!done && stop_status == 0) fallback = true;
    }
    if (FULL && stop_status != 0 && lane == 0) {   // no end cell: no walk, empty op string (R/wavefront_unialign.c:147-237)
      a.cigar_begin[pair] = a.cigar_off[pair + 1];
      a.cigar_len[pair] = 0;
    }
    if (split) {
      if (lane == 0) a.end_state[wi - w0] = make_int4(end_s, end_k, end_off, (fallback || stop_status != 0) ? 0 : 1);
    } else if (FULL && !SPLIT && !fallback && stop_status == 0) {
      // make this wave's history stores visible to its own loads
      __syncthreads();
      long long begin = 0;
      uint8_t* buf = a.cigar_ops + a.cigar_off[pair];
      if (!(a.debug & 1)) band_backtrace<NCH>(hist, a, plen, tlen, end_s, end_k, end_off, buf, &begin, lane, 64);
      if (lane == 0) {
        a.cigar_begin[pair] = a.cigar_off[pair] + begin;
        a.cigar_len[pair] = (int)((long long)plen + tlen - begin);
      }
    }
    if (a.done) __threadfence_system();   // the op bytes of every lane before the flag below
    if (lane == 0) {
      if (fallback) {
        a.status[pair] = WFA_INTERNAL_FALLBACK;
        if (a.fb_list) a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;   // (no list: the single-call path reads the status)
      } else if (stop_status != 0) {
        a.score[pair] = stop_score;
        a.status[pair] = stop_status;
      } else {
        a.score[pair] = result;
        a.status[pair] = 0;
      }
      if (a.done) { __threadfence_system(); __hip_atomic_store(&a.done[pair], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
}

template <int NCH, bool FULL, bool ADAPT, bool SEQLDS, bool PB, bool SPLIT, int X, int OE, int E, int OE2, int E2>
__global__ void __launch_bounds__(64)
wfa_band_kernel(const BandArgs a) {
  wfa_band_body<NCH, FULL, ADAPT, SEQLDS, PB, SPLIT, X, OE, E, OE2, E2>(a);
}
// the same body compiled for three waves per SIMD (<= 168 VGPRs): the 256-diagonal gap-affine-2p form, whose 180-odd
// registers would leave two
template <int NCH, bool FULL, bool ADAPT, bool SEQLDS, bool PB, bool SPLIT, int X, int OE, int E, int OE2, int E2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
wfa_band_kernel_w3(const BandArgs a) {
  wfa_band_body<NCH, FULL, ADAPT, SEQLDS, PB, SPLIT, X, OE, E, OE2, E2>(a);
}
template <int NCH, bool FULL, bool ADAPT, bool SEQLDS, bool PB, bool SPLIT, int X, int OE, int E, int OE2, int E2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
wfa_band_kernel_w4(const BandArgs a) {  // four waves per SIMD (<= 128 VGPRs): the 192-diagonal 2p form
  wfa_band_body<NCH, FULL, ADAPT, SEQLDS, PB, SPLIT, X, OE, E, OE2, E2>(a);
}

// Backtrace of a split launch: one THREAD per alignment, so that a wave keeps 64 dependent walks in flight.
// The walk emits run records into the top of the pair's own history slot; wfa_band_expand_kernel then
// writes the op bytes.
template <int NCH>
__global__ void __launch_bounds__(64)
wfa_band_bt_kernel(const BandArgs a) {
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
  if (t >= a.nwork) return;
  const int4 es = a.end_state[t];
  if (!es.w) return;
  const uint32_t wi = a.work_begin + t;
  const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
  const WfaPairMeta pm = a.meta[pair];
  int* hist = a.hist + (long long)t * a.hist_stride;
  long long begin = 0;
  int nruns = 0;
  band_backtrace<NCH>(hist, a, pm.plen, pm.tlen, es.x, es.y, es.z, nullptr, &begin, 0, 1,
                      reinterpret_cast<uint32_t*>(hist) + a.hist_stride - 1, &nruns);
  a.end_state[t] = make_int4((int)begin, nruns, 0, 2);
}

// Piggy-back form of the walk (the reference: R/wavefront_backtrace_offload.c, R/wavefront_pcigar.c:204-266).
// The history holds one byte of origin codes per (step, diagonal) — no offsets — so the walk is one byte load per
// hop and yields the edit events in reverse; the op string is then unpacked forwards, re-extending the matches after
// every event that lands in M against the packed sequences (a wavefront cell is always extended to its end, so the
// run after an event is the whole common prefix).  One thread per alignment.
__device__ __forceinline__ int pb_lcp(const uint32_t* __restrict__ P, const uint32_t* __restrict__ T, int v, int h, int maxn) {
  int n = 0;
  while (n < maxn) {
    const int vv = v + n, hh = h + n;
    const int pi = vv >> 4, ti = hh >> 4;
    const uint32_t p0 = P[pi], p1 = P[pi + 1], p2 = P[pi + 2], t0 = T[ti], t1 = T[ti + 1], t2 = T[ti + 2];
    const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)vv << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)hh << 1);
    const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)vv << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)hh << 1);
    int m = xl ? (__builtin_ctz(xl) >> 1) : (xh ? 16 + (__builtin_ctz(xh) >> 1) : 32);
    m = min(m, maxn - n);
    n += m;
    if (m < 32) break;
  }
  return n;
}

// Round 6: the forward unpack keeps a WINDOW of eight words of either sequence per lane in LDS ([word][lane]: no bank conflicts): the
// sequences are read forwards a few bases per event, so a refill (two 16-byte loads per sequence, for every active lane whenever any
// lane misses) serves several events.  (The back-walk is left on direct loads: windows of code rows fetch lines a walk never visits,
// DESIGN §8.)  -DWFA_WALK_WINDOWS=0: the direct loads of pb_lcp.
#ifndef WFA_WALK_WINDOWS
#define WFA_WALK_WINDOWS 1
#endif
template <int NCH>
__global__ void __launch_bounds__(64)
wfa_band_pb_bt_kernel(const BandArgs a) {
  const int W = a.seg_w ? a.seg_w : Band<NCH>::WI;   // (a.seg_w: the codes of wfa_seg_kernel<.., FULL>, records of seg_w bytes)
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
#if WFA_WALK_WINDOWS
  __shared__ uint32_t s_pw[8 * 64], s_tw[8 * 64];        // word j of the window: [j * 64 + lane]
  const int lane = threadIdx.x;
#endif
  if (t >= a.nwork) return;
  const int4 es = a.end_state[t];
  if (!es.w) return;
  const uint32_t wi = a.work_begin + t;
  const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
  const WfaPairMeta pm = a.meta[pair];
  const int plen = pm.plen, tlen = pm.tlen;
  int* hist = a.hist + (long long)t * a.hist_stride;
  const uint8_t* codes = reinterpret_cast<const uint8_t*>(hist);
  uint8_t* ev = reinterpret_cast<uint8_t*>(hist + a.pb_code_ints);
  uint32_t* runs = reinterpret_cast<uint32_t*>(hist + a.pb_code_ints + a.pb_event_ints);
  const int ev_cap = (int)min((long long)INT_MAX, a.pb_event_ints * 4);
  const int dx = a.x / a.g, doe = a.oe / a.g, de = a.e / a.g;
  const bool two = a.oe2 > 0;
  const int doe2 = two ? a.oe2 / a.g : 0, de2 = two ? a.e2 / a.g : 0;
  // ---- walk the codes back from the end cell (R/wavefront_backtrace.c:320-529 with the choices made at compute time)
  int si = es.x / a.g, k = es.y, comp = 0, nev = 0;
  // (round 5: the hop is selects, not branches — 64 walks per wave sit in different components and take different origins at every
  // hop; as an if / else ladder every path ran for every wave)
  if (two) {
    // comp: 0 M, 1 I1, 2 D1, 3 I2, 4 D2; src: 0 X, 1 D1, 2 D2, 3 I1, 4 I2; an event flagged 0x80 lands in M (a run of matches follows it)
    while (si > 0 && nev < ev_cap) {
      int cd = codes[(long long)si * W + (k & (W - 1))];
      if (a.pb_raw) {   // comparison bits of wfa_slim_kernel -> origin codes
        const int r = cd;
        const int mc = !(r & 128) ? 0 : !(r & 64) ? 2 : !(r & 32) ? 1 : !(r & 16) ? 4 : 3;
        cd = mc | ((r & 8) ? 0 : 8) | ((r & 4) ? 0 : 16) | ((r & 2) ? 0 : 32) | ((r & 1) ? 0 : 64);
      }
      const int src = (comp == 0) ? (cd & 7) : (int)((0x24130u >> (4 * comp)) & 7u);   // comp 1 -> I1 (3), 2 -> D1 (1), 3 -> I2 (4), 4 -> D2 (2)
      const bool is_d = src == 1 || src == 2, is_x = src == 0;
      // extension bit of the source component: I1 8, D1 16, I2 32, D2 64; the component it leads to when set; its lags
      const int ebit = (src == 3) ? 8 : (src == 1) ? 16 : (src == 4) ? 32 : 64;
      const bool ext = !is_x && (cd & ebit) != 0;
      const bool second = src == 2 || src == 4;
      const int lag = is_x ? dx : (ext ? (second ? de2 : de) : (second ? doe2 : doe));
      const int ncomp = ext ? ((src == 3) ? 1 : (src == 1) ? 2 : (src == 4) ? 3 : 4) : 0;
      ev[nev++] = (uint8_t)((is_x ? 'X' : (is_d ? 'D' : 'I')) | ((comp == 0) ? 0x80 : 0));
      k += is_x ? 0 : (is_d ? 1 : -1);
      si -= lag; comp = ncomp;
    }
  } else {
    // comp: 0 M, 1 I, 2 D; op: 0 X, 1 D, 2 I (the numbering of the origin code's low bits)
    while (si > 0 && nev < ev_cap) {
      int cd = codes[(long long)si * W + (k & (W - 1))];
      if (a.pb_raw) cd = (int)((0x2a6e195d084c084cull >> ((cd & 15) * 4)) & 15ull);   // comparison bits of wfa_slim_kernel -> origin codes
      const int op = (comp == 0) ? (cd & 3) : 3 - comp;
      const bool ext = op != 0 && ((cd >> (4 - op)) & 1) != 0;     // I: bit 2, D: bit 3
      ev[nev++] = (uint8_t)(((0x494458u >> (8 * op)) & 0xffu) | ((comp == 0) ? 0x80u : 0u));   // 'X', 'D', 'I'
      k += (op == 1) - (op == 2);
      si -= (op == 0) ? dx : (ext ? de : doe);
      comp = ext ? 3 - op : 0;
    }
  }
  // ---- unpack forwards from the cell of wavefront 0 on diagonal k (ends-free: offset max(k, 0), R/wavefront_aligner.c:259-302)
  const uint32_t* P = a.words + pm.p_woff;
  const uint32_t* T = a.words + pm.t_woff;
  int h = max(k, 0), v = h - k;
  int nruns = 0;
  long long total = 0;
  uint32_t run_op = 0, run_len = 0;   // the run being built stays in registers (round 5: the record was re-read from memory at every emit)
  auto emit = [&](int op, int n) {
    if (n <= 0) return;
    total += n;
    if (run_len > 0 && (int)run_op == op) { run_len += (uint32_t)n; return; }
    if (run_len > 0) runs[nruns++] = (run_len << 8) | run_op;
    run_op = (uint32_t)op; run_len = (uint32_t)n;
  };
#if WFA_WALK_WINDOWS
  int pw0 = -(1 << 20), tw0 = -(1 << 20);
  const int nwp_last = ((plen + 15) >> 4) + 1, nwt_last = ((tlen + 15) >> 4) + 1;   // (pb_lcp reads up to two words beyond a sequence: they exist)
  auto lcp = [&](int v_, int h_, int maxn) -> int {
    int n = 0;
    while (n < maxn) {
      const int vv = v_ + n, hh = h_ + n;
      const int pi = vv >> 4, ti = hh >> 4;
      const bool miss = !(pi >= pw0 && pi + 2 < pw0 + 8 && ti >= tw0 && ti + 2 < tw0 + 8);
      if (__any(miss)) {
        pw0 = pi; tw0 = ti;
        if (pi + 7 <= nwp_last && ti + 7 <= nwt_last) {
          typedef uint32_t walk_w4 __attribute__((ext_vector_type(4), aligned(4)));
          const walk_w4 pa = *reinterpret_cast<const walk_w4*>(P + pi), pb = *reinterpret_cast<const walk_w4*>(P + pi + 4);
          const walk_w4 ta = *reinterpret_cast<const walk_w4*>(T + ti), tb = *reinterpret_cast<const walk_w4*>(T + ti + 4);
          s_pw[0 * 64 + lane] = pa.x; s_pw[1 * 64 + lane] = pa.y; s_pw[2 * 64 + lane] = pa.z; s_pw[3 * 64 + lane] = pa.w;
          s_pw[4 * 64 + lane] = pb.x; s_pw[5 * 64 + lane] = pb.y; s_pw[6 * 64 + lane] = pb.z; s_pw[7 * 64 + lane] = pb.w;
          s_tw[0 * 64 + lane] = ta.x; s_tw[1 * 64 + lane] = ta.y; s_tw[2 * 64 + lane] = ta.z; s_tw[3 * 64 + lane] = ta.w;
          s_tw[4 * 64 + lane] = tb.x; s_tw[5 * 64 + lane] = tb.y; s_tw[6 * 64 + lane] = tb.z; s_tw[7 * 64 + lane] = tb.w;
        } else {   // (near a sequence's end word by word: nothing beyond its two look-ahead words is read)
#pragma unroll
          for (int j = 0; j < 8; ++j) { s_pw[j * 64 + lane] = P[min(pi + j, nwp_last)]; s_tw[j * 64 + lane] = T[min(ti + j, nwt_last)]; }
        }
      }
      const int po = pi - pw0, to = ti - tw0;
      const uint32_t p0 = s_pw[po * 64 + lane], p1 = s_pw[(po + 1) * 64 + lane], p2 = s_pw[(po + 2) * 64 + lane];
      const uint32_t t0 = s_tw[to * 64 + lane], t1 = s_tw[(to + 1) * 64 + lane], t2 = s_tw[(to + 2) * 64 + lane];
      const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)vv << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)hh << 1);
      const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)vv << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)hh << 1);
      int m = xl ? (__builtin_ctz(xl) >> 1) : (xh ? 16 + (__builtin_ctz(xh) >> 1) : 32);
      m = min(m, maxn - n);
      n += m;
      if (m < 32) break;
    }
    return n;
  };
#else
  auto lcp = [&](int v_, int h_, int maxn) -> int { return pb_lcp(P, T, v_, h_, maxn); };
#endif
  emit('I', h); emit('D', v);
  { const int n = lcp(v, h, min(plen - v, tlen - h)); emit('M', n); v += n; h += n; }
  for (int e = nev - 1; e >= 0; --e) {
    const int op = ev[e] & 0x7F;
    if (op == 'X') { emit('X', 1); ++v; ++h; }
    else if (op == 'I') { emit('I', 1); ++h; }
    else { emit('D', 1); ++v; }
    if (ev[e] & 0x80) { const int n = lcp(v, h, min(plen - v, tlen - h)); emit('M', n); v += n; h += n; }
  }
  emit('I', tlen - h); emit('D', plen - v);
  if (run_len > 0) runs[nruns++] = (run_len << 8) | run_op;
  a.end_state[t] = make_int4((int)((long long)plen + tlen - total), nruns, 1, 2);  // .z = 1: runs in forward order
}

#ifdef WFA_BAND_WALK_KERNELS  // the non-template kernels below are emitted by one translation unit (k_band.hip, index 5)
// One wave per alignment: run r (r = 0 is the LAST run of the op string) covers
// [end - sum(len[0..r]), end - sum(len[0..r-1])); lanes take 64 runs at a time, positions come from a wave
// prefix sum, short runs are written by their lane, long ones by the whole wave.
__global__ void __launch_bounds__(256)
wfa_band_expand_kernel(const BandArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t t = (blockIdx.x * 256u + threadIdx.x) >> 6;
  if (t >= a.nwork) return;
  const int4 es = a.end_state[t];
  if (es.w != 2) return;
  const uint32_t wi = a.work_begin + t;
  const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
  const WfaPairMeta pm = a.meta[pair];
  const uint32_t* top = reinterpret_cast<const uint32_t*>(a.hist + (long long)t * a.hist_stride) + a.hist_stride - 1;
  const bool fwd = es.z == 1;  // piggy-back unpack: runs in forward order, from the start of the op string
  const uint32_t* fruns = reinterpret_cast<const uint32_t*>(a.hist + (long long)t * a.hist_stride + a.pb_code_ints + a.pb_event_ints);
  uint8_t* buf = a.cigar_ops + a.cigar_off[pair];
  const int nruns = es.y;
  int end = pm.plen + pm.tlen;
  int start = es.x;
  for (int r0 = 0; r0 < nruns; r0 += 64) {
    const int r = r0 + lane;
    const uint32_t rec = (r < nruns) ? (fwd ? fruns[r] : *(top - r)) : 0u;
    const int len = (int)(rec >> 8);
    const uint8_t op = (uint8_t)(rec & 0xFFu);
    int cum = len;  // inclusive prefix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(cum, d, 64); if (lane >= d) cum += o; }
    const int pos = fwd ? start + cum - len : end - cum;  // first byte of this lane's run
    const bool is_long = len > 16;
    for (int i = 0; i < 16; ++i) if (!is_long && i < len) buf[pos + i] = op;
    unsigned long long lm = __ballot(is_long);
    while (lm) {
      const int l = __builtin_ctzll(lm);
      lm &= lm - 1;
      const int lpos = __builtin_amdgcn_readlane(pos, l), llen = __builtin_amdgcn_readlane(len, l);
      const uint8_t lop = (uint8_t)__builtin_amdgcn_readlane((int)op, l);
      for (int i = lane; i < llen; i += 64) buf[lpos + i] = lop;
    }
    end -= __builtin_amdgcn_readlane(cum, 63);
    start += __builtin_amdgcn_readlane(cum, 63);
  }
  if (lane == 0) {
    a.cigar_begin[pair] = a.cigar_off[pair] + es.x;
    a.cigar_len[pair] = pm.plen + pm.tlen - es.x;
  }
}

// Short reads (history of wfa_seg_kernel<.., FULL>): four alignments per wave, 16 lanes each.  The 16 lanes take 16
// runs at a time (positions from a prefix sum over the 16 lanes), then write run after run, 16 bytes per store.
__global__ void __launch_bounds__(256)
wfa_seg_expand_kernel(const BandArgs a) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, l = lane & 15;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  const uint32_t t = wave * 4u + (uint32_t)sub;
  int4 es = make_int4(0, 0, 0, 0);
  if (t < a.nwork) es = a.end_state[t];
  const bool live = es.w == 2;
  uint32_t pair = 0;
  WfaPairMeta pm; pm.p_woff = 0; pm.t_woff = 0; pm.plen = 0; pm.tlen = 0;
  if (live) {
    const uint32_t wi = a.work_begin + t;
    pair = a.worklist ? a.worklist[wi] : wi;
    pm = a.meta[pair];
  }
  // (round 4: the segments' history is piggy-back codes; the walk leaves its runs in forward order behind the codes and events)
  const uint32_t* fruns = reinterpret_cast<const uint32_t*>(a.hist + (long long)t * a.hist_stride + a.pb_code_ints + a.pb_event_ints);
  uint8_t* buf = live ? a.cigar_ops + a.cigar_off[pair] : nullptr;
  const int nruns = live ? es.y : 0;
  int start = es.x;
  int maxruns = nruns;  // the four alignments of the wave loop together
#pragma unroll
  for (int d = 16; d < 64; d <<= 1) maxruns = max(maxruns, __shfl_xor(maxruns, d, 64));
  for (int r0 = 0; r0 < maxruns; r0 += 16) {
    const int r = r0 + l;
    const uint32_t rec = (r < nruns) ? fruns[r] : 0u;
    const int len = (int)(rec >> 8);
    const int op = (int)(rec & 0xFFu);
    int cum = len;  // inclusive prefix sum over the 16 lanes of the alignment
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const int o = __shfl_up(cum, d, 16); if (l >= d) cum += o; }
    const int pos = start + cum - len;  // first byte of this lane's run
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int src = (lane & 48) + j;
      const int jpos = __shfl(pos, src, 64), jlen = __shfl(len, src, 64), jop = __shfl(op, src, 64);
      for (int i = l; i < jlen; i += 16) buf[jpos + i] = (uint8_t)jop;
    }
    start += __shfl(cum, (lane & 48) + 15, 64);
  }
  if (live && l == 0) {
    a.cigar_begin[pair] = a.cigar_off[pair] + es.x;
    a.cigar_len[pair] = pm.plen + pm.tlen - es.x;
  }
}

// Short reads, history of wfa_lane_kernel<.., FULL> (wfa_lane.hpp): one THREAD per alignment walks the comparison bits back from
// the end cell (R/wavefront_backtrace.c:320-529 with the choices made at compute time: bit 3 of a slot's nibble: the mismatch
// candidate is below the best gap candidate; bit 2: deletion below insertion; bit 1 / 0: the extension of I / D is below its
// opening), then unpacks forwards from the cell (score 0, offset 0), re-extending the matches on the packed words, into run
// records {length << 8 | op} at the start of the alignment's slot.  An alignment whose edits do not fit (more than 31 runs) is
// handed on to the next stage.
__global__ void __launch_bounds__(64)
wfa_lane_walk_kernel(const BandArgs a) {
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
  if (t >= a.nwork) return;
  const int4 es = a.end_state[t];
  if (es.w != 1) return;
  const uint32_t wi = a.work_begin + t;
  const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
  const WfaPairMeta pm = a.meta[pair];
  const int plen = pm.plen, tlen = pm.tlen;
  const int dx = a.x / a.g, doe = a.oe / a.g, de = a.e / a.g;
  const int ln = es.y & 0xff, t_end = es.z;
  const uint2* const rec_end = a.lane_codes + (unsigned long long)(unsigned)es.x * 64ull + (unsigned)ln;
  // events, last edit first: 4 bits each (op: 1 X, 2 I, 3 D; bit 3: lands in M) in two 64-bit words
  unsigned long long ev_lo = 0ull, ev_hi = 0ull;
  int tt = t_end, j = es.y >> 8, comp = 0, nev = 0;
  // (the records of the next CH steps down are requested together — independent loads, one round trip — and the hops that land on
  // them are taken from registers: 2-4 round trips per alignment instead of one per edit)
  constexpr int CH = 8;
  while (tt > 0 && nev < 32) {
    const int top = tt;
    uint2 cc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int st = top - i;
      cc[i] = (st > 0) ? rec_end[-(long long)(t_end - st) * 64ll] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (tt == top - i && tt > 0 && nev < 32) {
        const uint2 c = cc[i];
        const uint32_t nib = (((j < 8) ? c.x : c.y) >> (4 * (j & 7))) & 0xFu;
        uint32_t ev;
        if (comp == 0) {
          if (!(nib & 8u)) { ev = 1u | 8u; tt -= dx; }
          else if (!(nib & 4u)) { ev = 3u | 8u; ++j; if (!(nib & 1u)) { tt -= de; comp = 2; } else tt -= doe; }
          else { ev = 2u | 8u; --j; if (!(nib & 2u)) { tt -= de; comp = 1; } else tt -= doe; }
        } else if (comp == 1) {
          ev = 2u; --j;
          if (!(nib & 2u)) tt -= de; else { tt -= doe; comp = 0; }
        } else {
          ev = 3u; ++j;
          if (!(nib & 1u)) tt -= de; else { tt -= doe; comp = 0; }
        }
        if (nev < 16) ev_lo |= (unsigned long long)ev << (4 * nev); else ev_hi |= (unsigned long long)ev << (4 * (nev - 16));
        ++nev;
      }
    }
  }
  uint32_t* const runs = reinterpret_cast<uint32_t*>(a.hist + (long long)t * a.hist_stride);
  const int max_runs = (int)a.hist_stride;
  bool over = (tt != 0);
  int nruns = 0, total = 0;
  if (!over) {
    const uint32_t* P = a.words + pm.p_woff;
    const uint32_t* T = a.words + pm.t_woff;
    int v = 0, h = 0;
    uint32_t cur_op = 'M';
    int cur_len = 0;
    auto emit = [&](uint32_t op, int n) {
      if (n <= 0) return;
      total += n;
      if (op == cur_op) { cur_len += n; return; }
      if (cur_len > 0) { if (nruns < max_runs) runs[nruns] = ((uint32_t)cur_len << 8) | cur_op; ++nruns; }
      cur_op = op; cur_len = n;
    };
    { const int n = pb_lcp(P, T, v, h, min(plen - v, tlen - h)); emit('M', n); v += n; h += n; }
    for (int e = nev - 1; e >= 0; --e) {
      const uint32_t ev = (uint32_t)(((e < 16) ? (ev_lo >> (4 * e)) : (ev_hi >> (4 * (e - 16)))) & 0xFull);
      const uint32_t op = ev & 7u;
      if (op == 1u) { emit('X', 1); ++v; ++h; }
      else if (op == 2u) { emit('I', 1); ++h; }
      else { emit('D', 1); ++v; }
      if (ev & 8u) { const int n = pb_lcp(P, T, v, h, min(plen - v, tlen - h)); emit('M', n); v += n; h += n; }
    }
    emit('I', tlen - h); emit('D', plen - v);
    if (cur_len > 0) { if (nruns < max_runs) runs[nruns] = ((uint32_t)cur_len << 8) | cur_op; ++nruns; }
    over = nruns > max_runs;
  }
  if (over) {
    a.end_state[t] = make_int4(0, 0, 0, 0);
    a.status[pair] = WFA_INTERNAL_FALLBACK;
    a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;
  } else {
    a.end_state[t] = make_int4(plen + tlen - total, nruns, 1, 2);   // {first op, runs, forward order, ready}
  }
}

// The op bytes from the run records of wfa_lane_walk_kernel.  Four alignments per wave, 16 lanes each.
__global__ void __launch_bounds__(256)
wfa_lane_expand_kernel(const BandArgs a) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, l = lane & 15;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  const uint32_t t = wave * 4u + (uint32_t)sub;
  int4 es = make_int4(0, 0, 0, 0);
  if (t < a.nwork) es = a.end_state[t];
  const bool live = es.w == 2;
  uint32_t pair = 0;
  WfaPairMeta pm; pm.p_woff = 0; pm.t_woff = 0; pm.plen = 0; pm.tlen = 0;
  if (live) {
    const uint32_t wi = a.work_begin + t;
    pair = a.worklist ? a.worklist[wi] : wi;
    pm = a.meta[pair];
  }
  const uint32_t* runs = reinterpret_cast<const uint32_t*>(a.hist + (long long)t * a.hist_stride);
  uint8_t* buf = live ? a.cigar_ops + a.cigar_off[pair] : nullptr;
  const int nruns = live ? es.y : 0;
  int start = es.x;
  int maxruns = nruns;  // the four alignments of the wave loop together
#pragma unroll
  for (int d = 16; d < 64; d <<= 1) maxruns = max(maxruns, __shfl_xor(maxruns, d, 64));
  for (int r0 = 0; r0 < maxruns; r0 += 16) {
    const int r = r0 + l;
    const uint32_t rec = (r < nruns) ? runs[r] : 0u;
    const int len = (int)(rec >> 8);
    const int op = (int)(rec & 0xFFu);
    int cum = len;  // inclusive prefix sum over the 16 lanes of the alignment
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const int o = __shfl_up(cum, d, 16); if (l >= d) cum += o; }
    const int pos = start + cum - len;  // first byte of this lane's run
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int src = (lane & 48) + j;
      const int jpos = __shfl(pos, src, 64), jlen = __shfl(len, src, 64), jop = __shfl(op, src, 64);
      for (int i = l; i < jlen; i += 16) buf[jpos + i] = (uint8_t)jop;
    }
    start += __shfl(cum, (lane & 48) + 15, 64);
  }
  if (live && l == 0) {
    a.cigar_begin[pair] = a.cigar_off[pair] + es.x;
    a.cigar_len[pair] = pm.plen + pm.tlen - es.x;
  }
}

// (the walk can hand a pair on, so it runs before the next stage; the expand only reads run records and writes op bytes, so
// the host may put it on a side stream under the stages behind the lane kernel)
inline int launch_lane_expand_impl(const BandArgs& a, hipStream_t walk_stream, hipStream_t expand_stream) {
  if (a.nwork == 0) return 0;
  if (walk_stream) {
    hipLaunchKernelGGL(wfa_lane_walk_kernel, dim3((a.nwork + 63u) / 64u), dim3(64), 0, walk_stream, a);
    if (hipGetLastError() != hipSuccess) return -1;
  }
  if (expand_stream) {
    hipLaunchKernelGGL(wfa_lane_expand_kernel, dim3((a.nwork + 15u) / 16u), dim3(256), 0, expand_stream, a);
    if (hipGetLastError() != hipSuccess) return -1;
  }
  return 0;
}

inline int launch_band_bt_impl(const BandArgs& a, int nch, hipStream_t stream) {
  const unsigned grid = (a.nwork + 63u) / 64u;
  if (grid == 0) return 0;
  if (a.pb) {
    if (nch == 1) hipLaunchKernelGGL((wfa_band_pb_bt_kernel<1>), dim3(grid), dim3(64), 0, stream, a);
    else if (nch == 2) hipLaunchKernelGGL((wfa_band_pb_bt_kernel<2>), dim3(grid), dim3(64), 0, stream, a);
    else if (nch == 3) hipLaunchKernelGGL((wfa_band_pb_bt_kernel<3>), dim3(grid), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((wfa_band_pb_bt_kernel<4>), dim3(grid), dim3(64), 0, stream, a);
  } else if (nch == 1) hipLaunchKernelGGL((wfa_band_bt_kernel<1>), dim3(grid), dim3(64), 0, stream, a);
  else if (nch == 3) hipLaunchKernelGGL((wfa_band_bt_kernel<3>), dim3(grid), dim3(64), 0, stream, a);
  else if (nch == 2) hipLaunchKernelGGL((wfa_band_bt_kernel<2>), dim3(grid), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((wfa_band_bt_kernel<4>), dim3(grid), dim3(64), 0, stream, a);
  if (hipGetLastError() != hipSuccess) return -1;
  if (a.seg_w) hipLaunchKernelGGL(wfa_seg_expand_kernel, dim3((a.nwork + 15u) / 16u), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(wfa_band_expand_kernel, dim3((a.nwork + 3u) / 4u), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

#endif  // WFA_BAND_WALK_KERNELS

#ifndef __HIPCC_RTC__   // ---- host side (launch code, shape tables) ----
template <int NCH, bool FULL, bool ADAPT, bool PB, bool SPLIT, int X, int OE, int E, int OE2, int E2>
static int launch_band_k(const BandArgs& a, bool seqlds, long long grid, hipStream_t stream) {
  const size_t smem = seqlds ? (size_t)a.lds_words * 2 * sizeof(uint32_t) : 0;
  // (the piggy-back forms need a few more registers: compiled without the occupancy cap rather than with spills)
  if constexpr (OE2 > 0 && NCH == 3) {
    if (seqlds) hipLaunchKernelGGL((wfa_band_kernel_w4<NCH, FULL, ADAPT, true, PB, SPLIT, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else hipLaunchKernelGGL((wfa_band_kernel_w4<NCH, FULL, ADAPT, false, PB, SPLIT, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), 0, stream, a);
  } else if constexpr (OE2 > 0 && NCH == 4 && !PB) {
    if (seqlds) hipLaunchKernelGGL((wfa_band_kernel_w3<NCH, FULL, ADAPT, true, PB, SPLIT, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else hipLaunchKernelGGL((wfa_band_kernel_w3<NCH, FULL, ADAPT, false, PB, SPLIT, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), 0, stream, a);
  } else {
    if (seqlds) hipLaunchKernelGGL((wfa_band_kernel<NCH, FULL, ADAPT, true, PB, SPLIT, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else hipLaunchKernelGGL((wfa_band_kernel<NCH, FULL, ADAPT, false, PB, SPLIT, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int NCH, bool FULL, bool ADAPT, int X, int OE, int E, int OE2, int E2>
static int launch_band_t(const BandArgs& a, bool seqlds, long long grid, hipStream_t stream) {
  constexpr bool CAN_PB = FULL;
  if (FULL && a.split) {  // history slot per pair, walk in its own kernel; piggy-back history on request
    if (CAN_PB && a.pb) return launch_band_k<NCH, FULL, ADAPT, CAN_PB, FULL, X, OE, E, OE2, E2>(a, seqlds, grid, stream);
    return launch_band_k<NCH, FULL, ADAPT, false, FULL, X, OE, E, OE2, E2>(a, seqlds, grid, stream);
  }
  return launch_band_k<NCH, FULL, ADAPT, false, false, X, OE, E, OE2, E2>(a, seqlds, grid, stream);
}

template <int X, int OE, int E, int OE2, int E2>
static int launch_band_shape(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds, long long grid, hipStream_t stream) {
#define WFA_BAND_CASE(N)                                                                                       \
  if (nch == N) {                                                                                              \
    if (full) return adapt ? launch_band_t<N, true, true, X, OE, E, OE2, E2>(a, seqlds, grid, stream)          \
                           : launch_band_t<N, true, false, X, OE, E, OE2, E2>(a, seqlds, grid, stream);        \
    return adapt ? launch_band_t<N, false, true, X, OE, E, OE2, E2>(a, seqlds, grid, stream)                   \
                 : launch_band_t<N, false, false, X, OE, E, OE2, E2>(a, seqlds, grid, stream);                 \
  }
  WFA_BAND_CASE(1)
  WFA_BAND_CASE(2)
  WFA_BAND_CASE(4)
  if constexpr (OE2 > 0) { WFA_BAND_CASE(3) }  // 192 diagonals: what most gap-affine-2p wavefronts need
#undef WFA_BAND_CASE
  return -1;
}

// ---- host entry points.  Every penalty shape is compiled in its own translation unit (csrc/k_band.hip, once per
// index; index 4 = gap-affine-2p, index 5 = the walks / expansion kernels) so that the library builds in parallel.
int launch_band_bt(const BandArgs& a, int nch, hipStream_t stream);
int launch_lane_expand(const BandArgs& a, hipStream_t walk_stream, hipStream_t expand_stream);   // walk the codes of wfa_lane_kernel<.., FULL> into run records / run records into op bytes (either stream may be null: that half is skipped)
#define WFA_BAND_DECL(i, x, oe, e) int launch_band_s##i(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds, long long grid, hipStream_t stream);
WFA_BAND_SHAPES(WFA_BAND_DECL)
#undef WFA_BAND_DECL
int launch_band_s4(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds, long long grid, hipStream_t stream);  // 2p
// the slim form (wfa_slim.hpp, csrc/k_slim.hip): one translation unit per gap-affine shape
#define WFA_SLIM_DECL(i, x, oe, e) int launch_slim_s##i(const BandArgs& a, int nch, bool full, long long grid, hipStream_t stream); \
  int launch_slim_mailbox_s##i(const BandArgs& a, bool full, hipStream_t stream, SlimMailbox* mb);
WFA_BAND_SHAPES(WFA_SLIM_DECL)
#undef WFA_SLIM_DECL
int launch_slim_s4(const BandArgs& a, int nch, bool full, long long grid, hipStream_t stream);  // 2p
// launches the slim form covers: the wf-adaptive long-read cascade — its first window (128 diagonals, gap-affine-2p: 192; score-only
// or the piggy-back history of a split launch) and the 256-diagonal stage behind it (score-only or the explicit int16 history, walked
// in-kernel)
inline bool slim_takes(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds) {
  if (!a.slim || !seqlds || a.debug != 0) return false;
  if (adapt ? a.heur != 1 : a.heur != 0) return false;                          // wf-adaptive or no heuristic (X-drop: wfa_band_kernel)
  if (nch == (a.oe2 > 0 ? 3 : 2)) return !full || (a.split ? a.pb != 0 : a.h16 != 0);   // piggy-back slots, or the explicit int16 history walked in-kernel
  if (a.win) return nch == 4 && a.oe2 == 0 && (!full || (a.split && a.pb != 0));   // (windowed sequences: the 256-diagonal first stage of reads over 26 kb)
  if (nch == 4) return !full || (a.split ? (a.pb != 0 && a.oe2 == 0) : a.h16 != 0);   // (split: gap-affine only — the 2p form of 256 diagonals would take ~300 registers)
  return false;
}

// configurations the band kernel covers: gap-affine / gap-affine-2p with an instantiated penalty shape
// (x, o1 + e1, e1 [, o2 + e2, e2]) / gcd; 2p: pywfa's default 4/6/2/24/1
#define WFA_BAND_SHAPES_2P(F) F(4, 8, 2, 25, 1)
inline int band_gcd(const WfaDevConfig& c, bool two) {
  int g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  if (two) g = gcd_int(gcd_int(g, c.o2 + c.e2), c.e2);
  return g;
}
// none / wf-adaptive / X-drop, with or without a step limit
inline bool band_supported(const WfaDevConfig& c, int ncomp) {
  if ((ncomp != 3 && ncomp != 5) || c.match != 0 || c.wildcard >= 0) return false;
  if (c.heuristic < 0 || c.heuristic > 2) return false;
  const bool two = ncomp == 5;
  const int g = band_gcd(c, two);
  const int X = c.x / g, OE = (c.o1 + c.e1) / g, E = c.e1 / g;
  if (two) {
    const int OE2 = (c.o2 + c.e2) / g, E2 = c.e2 / g;
#define WFA_BAND_MATCH2(x, oe, e, oe2, e2) if (X == x && OE == oe && E == e && OE2 == oe2 && E2 == e2) return true;
    WFA_BAND_SHAPES_2P(WFA_BAND_MATCH2)
#undef WFA_BAND_MATCH2
    return c.rtc && rtc_shape_ok(X, OE, E, OE2, E2) && rtc_available();
  }
#define WFA_BAND_MATCH(i, x, oe, e) if (X == x && OE == oe && E == e) return true;
  WFA_BAND_SHAPES(WFA_BAND_MATCH)
#undef WFA_BAND_MATCH
  return c.rtc && rtc_shape_ok(X, OE, E) && rtc_available();
}

// the banded kernel of a shape without an instantiation, compiled at run time (csrc/wfa_rtc.cpp): the same choice of kernel and
// template arguments launch_band_shape / _t / _k make at compile time
inline int launch_band_rtc(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds, long long grid, hipStream_t stream) {
  const int g = a.g, X = a.x / g, OE = a.oe / g, E = a.e / g, OE2 = a.oe2 > 0 ? a.oe2 / g : 0, E2 = a.oe2 > 0 ? a.e2 / g : 0;
  if (!(nch == 1 || nch == 2 || nch == 4 || (nch == 3 && OE2 > 0))) return -1;
  const bool split = full && a.split, pb = split && a.pb;
  const char* kernel = (OE2 > 0 && nch == 3) ? "wfa_band_kernel_w4" : (OE2 > 0 && nch == 4 && !pb) ? "wfa_band_kernel_w3" : "wfa_band_kernel";
  const std::string name = std::string("wfa::") + kernel + "<" + std::to_string(nch) + ", " + rtc_bool(full) + ", " + rtc_bool(adapt) + ", " + rtc_bool(seqlds) + ", " +
                           rtc_bool(pb) + ", " + rtc_bool(split) + ", " + std::to_string(X) + ", " + std::to_string(OE) + ", " + std::to_string(E) + ", " +
                           std::to_string(OE2) + ", " + std::to_string(E2) + ">";
  const size_t smem = seqlds ? (size_t)a.lds_words * 2 * sizeof(uint32_t) : 0;
  return rtc_launch("wfa_band.hpp", name, (unsigned)grid, 64, smem, stream, &a, sizeof(a));
}

// shapes without an instantiation whose slim form is compiled at run time: gap-affine with rings of up to twelve steps (round 5) and,
// since the deep M history of gap-affine-2p lives in an LDS ring (round 6: eleven ring registers per chunk whatever o2 + e2 is),
// gap-affine-2p with x < o + e < o2 + e2 and a ring of up to 40 rows; the rest takes wfa_band_kernel's run-time form as before
inline bool slim_rtc_shape_ok(int X, int OE, int E, int OE2, int E2 = 0) {
  if (OE2 == 0) return (X > OE ? X : OE) <= 12 && E <= 3;
  return X < OE && OE < OE2 && X <= 8 && E <= 3 && E2 >= 1 && E2 <= 2 && OE2 - X <= 40;
}
inline int launch_slim_rtc(const BandArgs& a, int nch, bool full, long long grid, hipStream_t stream) {
  const int g = a.g, X = a.x / g, OE = a.oe / g, E = a.e / g, OE2 = a.oe2 / g, E2 = a.e2 / g;
  const int hist = (full && a.split) ? 1 : full ? 2 : 0;
  size_t smem = (size_t)a.lds_words * 2 * sizeof(uint32_t);
  std::string name;
  if (OE2 > 0) {
    name = std::string("wfa::") + (nch == 4 ? "wfa_slim_kernel_tail<4, " : "wfa_slim_kernel_2p<3, ") + std::to_string(hist) + ", " + std::to_string(X) + ", " +
           std::to_string(OE) + ", " + std::to_string(E) + ", " + std::to_string(OE2) + ", " + std::to_string(E2) + ">";
    smem += (size_t)(OE2 - X) * (size_t)(64 * (nch == 4 ? 4 : 3) + 2) * sizeof(short);   // the LDS ring of the deep M history (wfa_slim.hpp)
  } else {
    name = std::string("wfa::") + (nch == 4 ? "wfa_slim_kernel_tail<4, " : "wfa_slim_kernel<2, ") + std::to_string(hist) + ", " + std::to_string(X) + ", " +
           std::to_string(OE) + ", " + std::to_string(E) + (a.win ? ", 0, 0, true>" : ", 0, 0>");
  }
  return rtc_launch("wfa_slim.hpp", name, (unsigned)grid, 64, smem, stream, &a, sizeof(a));
}

// true when launch_band() sends this launch to wfa_slim_kernel: slim_takes() and a shape the library has an instantiation of or
// compiles the slim form of (the host sets BandArgs::pb_raw from it before the launch: the walk must know which kernel wrote the codes)
inline bool slim_launches(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds) {
  if (!slim_takes(a, nch, full, adapt, seqlds) || (rtc_force_all() && rtc_available())) return false;
  const int g = a.g, X = a.x / g, OE = a.oe / g, E = a.e / g;
  if (a.oe2 > 0) {
    const int OE2 = a.oe2 / g, E2 = a.e2 / g;
#define WFA_SLIM_MATCH2(x_, oe_, e_, oe2_, e2_) if (X == x_ && OE == oe_ && E == e_ && OE2 == oe2_ && E2 == e2_) return true;
    WFA_BAND_SHAPES_2P(WFA_SLIM_MATCH2)
#undef WFA_SLIM_MATCH2
    return slim_rtc_shape_ok(X, OE, E, OE2, E2) && rtc_available();
  }
#define WFA_SLIM_MATCH(i, x, oe, e) if (X == x && OE == oe && E == e) return true;
  WFA_BAND_SHAPES(WFA_SLIM_MATCH)
#undef WFA_SLIM_MATCH
  return slim_rtc_shape_ok(X, OE, E, 0) && rtc_available();
}

// an instance of the resident one-pair kernel (wfa_slim.hpp) for this configuration; -1: none (gap-affine-2p, a shape without an
// instantiation, a launch the slim form does not take)
inline int launch_slim_mailbox(const BandArgs& a, bool full, bool adapt, hipStream_t stream, SlimMailbox* mb) {
  if (a.oe2 > 0 || a.split || !slim_takes(a, 2, full, adapt, true) || (rtc_force_all() && rtc_available())) return -1;
  const int g = a.g, X = a.x / g, OE = a.oe / g, E = a.e / g;
#define WFA_SLIM_MB(i, x, oe, e) if (X == x && OE == oe && E == e) return launch_slim_mailbox_s##i(a, full, stream, mb);
  WFA_BAND_SHAPES(WFA_SLIM_MB)
#undef WFA_SLIM_MB
  return -1;
}

inline int launch_band(const BandArgs& a, int nch, bool full, bool adapt, bool seqlds, long long grid, hipStream_t stream) {
  const int g = a.g, X = a.x / g, OE = a.oe / g, E = a.e / g;
  if (rtc_force_all() && rtc_available()) return launch_band_rtc(a, nch, full, adapt, seqlds, grid, stream);
  if (a.oe2 > 0) {
    const int OE2 = a.oe2 / g, E2 = a.e2 / g;
#define WFA_BAND_LAUNCH2(x_, oe_, e_, oe2_, e2_) if (X == x_ && OE == oe_ && E == e_ && OE2 == oe2_ && E2 == e2_) return slim_takes(a, nch, full, adapt, seqlds) ? launch_slim_s4(a, nch, full, grid, stream) : launch_band_s4(a, nch, full, adapt, seqlds, grid, stream);
    WFA_BAND_SHAPES_2P(WFA_BAND_LAUNCH2)
#undef WFA_BAND_LAUNCH2
    if (slim_takes(a, nch, full, adapt, seqlds) && slim_rtc_shape_ok(X, OE, E, OE2, E2)) return launch_slim_rtc(a, nch, full, grid, stream);
    return launch_band_rtc(a, nch, full, adapt, seqlds, grid, stream);
  }
#define WFA_BAND_LAUNCH(i, x, oe, e) if (X == x && OE == oe && E == e) return slim_takes(a, nch, full, adapt, seqlds) ? launch_slim_s##i(a, nch, full, grid, stream) : launch_band_s##i(a, nch, full, adapt, seqlds, grid, stream);
  WFA_BAND_SHAPES(WFA_BAND_LAUNCH)
#undef WFA_BAND_LAUNCH
  if (slim_takes(a, nch, full, adapt, seqlds) && slim_rtc_shape_ok(X, OE, E, 0)) return launch_slim_rtc(a, nch, full, grid, stream);
  return launch_band_rtc(a, nch, full, adapt, seqlds, grid, stream);
}

#endif  // __HIPCC_RTC__

}  // namespace wfa
