// wfa_slim.hpp — the slim form of the banded kernel (wfa_band.hpp): the step of the long-read hot paths (BASELINE C3: 10 kb reads,
// gap-affine, wf-adaptive, full CIGAR; C4 with wf-adaptive: gap-affine-2p, ends-free) rebuilt around its instruction count.  Same
// algorithm, same sliding window (128 diagonals for gap-affine, 192 for gap-affine-2p), same piggy-back history layout and the same
// results as wfa_band_kernel<NCH, FULL, true, true, FULL, FULL, X, OE, E, OE2, E2>; what differs is what a wave issues per score
// step (R = /root/reference/pywfa/WFA2_lib/wavefront):
//   * offsets are kept DOUBLED (2 x h): an offset is then the bit position of its base in the 2-bit packed text, the extension
//     (R/wavefront_extend_kernels.c:64-110) needs no conversion on the way in or out, and every comparison of the step is
//     invariant under the scaling;
//   * no divergent branch and no early exit inside the step: every branch of the loop is a scalar branch and the loops have one
//     exit each, so the compiler keeps the control flow as written (one lane-dependent `if`, or a `break`, makes it linearise the
//     whole loop body behind flag registers);
//   * the step runs on the ACTIVE chunks only — 1 .. NCH chunks of 64 diagonals, chosen once per block of 8 steps at the hull check
//     — as one straight body per count: the 2p form of wfa_band_kernel always walks its 192 diagonals although the wavefront a
//     cut-off keeps is a few dozen diagonals wide once the free begin has been trimmed;
//   * the wf-adaptive cut-off (R/wavefront_heuristic.c:176-293) takes its wave minimum in six DPP steps, its limits in window
//     positions from 64-bit masks, and drops lanes through a mask in each value's own register;
//   * termination (R/wavefront_termination.c:37-61, 115-162) is one compare per chunk against a per-lane threshold that folds the
//     end-to-end cell and both ends-free borders;
//   * the k-1 / k+1 neighbours (R/wavefront_compute_affine.c:44-86, R/wavefront_compute_affine2p.c:45-106) of the one-chunk form
//     come through registers whose edge lane is NULL for good, so a shift is one DPP move;
//   * one piggy-back byte per ACTIVE diagonal and step: the sign bits of the candidates' differences as they fall out of the
//     subtractions (BandArgs::pb_raw; the walk decodes them).
// Scope: gap-affine / gap-affine-2p with an instantiated shape, match = 0, wf-adaptive or no heuristic, end-to-end or ends-free,
// sequences staged in LDS (reads <= 10 kb); score-only, the piggy-back history of a split launch, or the explicit history of an
// unsplit one (walked in-kernel; the single-call path included).  Everything else (X-drop, longer reads, run-time shapes) stays with
// wfa_band_kernel; a pair whose window overflows is handed on exactly as there.
#pragma once
#include "wfa_band.hpp"

#ifndef WFA_SLIM_HP
#define WFA_SLIM_HP 8
#endif
#ifndef WFA_SLIM_WAVES
#define WFA_SLIM_WAVES 7
#endif
#ifndef WFA_SLIM_WAVES_2P
#define WFA_SLIM_WAVES_2P 3
#endif

namespace wfa {

// wave-wide minimum: butterflies inside the rows of 16 lanes, then row_bcast:15 / row_bcast:31 carry it to lane 63
__device__ __forceinline__ int slim_wave_min(int x) {
  int v;
  asm("s_nop 1\n\t"
      "v_min_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "=&v"(v) : "v"(x));
  return __builtin_amdgcn_readlane(v, 63);
}

// r = (lane's bit of mask) ? r : NULL, in r's own register (a plain select makes the compiler keep both values alive and copy
// registers where the paths of the step join)
__device__ __forceinline__ void slim_keep(int& r, unsigned long long mask) {
  asm("v_cndmask_b32 %0, -2.0, %0, %1" : "+v"(r) : "s"(mask));   // (-2.0 = 0xC0000000 = WFA_OFFSET_NULL)
}

// one probe of the extension on doubled coordinates: v2 / h2 = bit positions in the packed pattern / text; returns the number of
// matching BITS (even; >= 64: none of the 32 bases differs)
__device__ __forceinline__ uint32_t slim_probe(const uint32_t* sP, const uint32_t* sT, int v2, int h2, int pwb = 0, int twb = 0) {
  const int pi = (v2 >> 5) - pwb, ti = (h2 >> 5) - twb;   // (pwb / twb: first word of the staged window, 0 unless WIN)
  const uint32_t p0 = sP[pi], p1 = sP[pi + 1], p2 = sP[pi + 2], t0 = sT[ti], t1 = sT[ti + 1], t2 = sT[ti + 2];
  const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v2) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h2);
  const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v2) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h2);
  return min(band_ffbl(xl), band_ffbl(xh) | 32u) & ~1u;   // (v_ffbl_b32 gives ~0 for 0)
}

// lanes of chunk c (window positions 64c .. 64c + 63) inside [lo, hi]
__device__ __forceinline__ unsigned long long slim_range_mask(int c, int lo, int hi) {
  const int l = lo - 64 * c, h = hi - 64 * c;
  if (l > 63 || h < 0) return 0ull;
  return (~0ull << max(l, 0)) & (~0ull >> (63 - min(h, 63)));
}

// HIST: 0 score only; 1 piggy-back history of a split launch (one byte per active diagonal and step in the pair's slot; the walk runs
// in wfa_band_pb_bt_kernel afterwards); 2 the explicit history of wfa_band_kernel's unsplit form ({M, I, D, window base} per window
// position and step in the workgroup's slice, walked in-kernel by band_backtrace) — the stage BEHIND the split one, which takes the
// few pairs a window hands on: there the latency of one alignment is what counts, and this step is a third of wfa_band_kernel's.
// WIN (round 5): reads too long for LDS — a WINDOW of a.lds_words words of each sequence is staged and moved along as the alignment
// advances.  Offsets never decrease from one wavefront to the next (M, I: h + 1; D: the same h on the diagonal below, where v = h - k
// is one more), so the smallest position of a live diagonal is a lower bound for every later read; the extension tests before each
// round of probes whether a live diagonal is about to read past the window's end and then re-stages both windows from the smallest
// live positions (a pair whose live diagonals span more than a window is handed on).
template <int NCH, int HIST, int X, int OE, int E, int OE2, int E2, bool WIN = false>
__device__ __forceinline__ void wfa_slim_body(const BandArgs& a, const uint32_t* __restrict__ inl = nullptr, int* __restrict__ res2 = nullptr) {
  // (res2: the resident one-pair kernel takes the pair's {score, status} in registers as well: its score-only answer is one store)
  constexpr bool FULL = HIST != 0, PBH = HIST == 1, XH = HIST == 2;
  static_assert(!WIN || OE2 == 0, "the windowed form: gap-affine");
  constexpr bool TWO = OE2 > 0;  // gap-affine-2p: second pair of gap components (R/wavefront_compute_affine2p.c:45-106)
  constexpr int E2D = TWO ? E2 : 1;
  typedef Band<NCH> BD;
  constexpr int W = BD::W, WI = BD::WI, HP = WFA_SLIM_HP;   // steps between hull checks
  // M history in registers: depths 1 .. DM.  gap-affine: max(x, o + e).  gap-affine-2p (round 6): depths 1 .. x only — the deeper
  // history (depths x + 1 .. o2 + e2, read on the neighbour diagonals at depths o + e and o2 + e2) lives in an LDS ring of
  // RR = o2 + e2 - x rows of int16 (doubled) offsets indexed by window position, one pad cell at either end that is NULL for good:
  // a neighbour read is a ds_read_i16 at a constant byte offset, the ring needs no register moves, and 14 of the 25 ring registers
  // per chunk (and every spill: round 5 ran 256 B of scratch per lane) are gone.  A row is written when its value leaves the
  // registers (step s writes M[s - x] over M[s - o2 - e2], which the previous step read last); NULL is stored as -32768 and read back
  // as such — a dead gap cell can then be a small negative number instead of WFA_OFFSET_NULL, which every test of the step treats
  // alike (>= 0 is alive); so that such values cannot creep up to 0 (+2 per step at most) the gap rings are cleaned at every hull
  // check, and M itself is kept exact by the clamp of compute-next.
  constexpr bool RING = TWO;
  constexpr int DM = RING ? X : ((X > OE) ? X : OE);
  constexpr int NP = 1;
  constexpr int RR = RING ? OE2 - X : 1;                     // rows of the LDS ring
  static_assert(!TWO || (OE > X && OE2 > OE), "2p: x < o + e < o2 + e2 (the ring holds the depths beyond x)");
  constexpr int NUL = WFA_OFFSET_NULL;
  extern __shared__ uint32_t slds[];
  uint32_t* const sP = slds;
  uint32_t* const sT = slds + a.lds_words;
  constexpr int RW = 64 * NCH + 2;                           // cells of a ring row: [pad][window positions][pad]
  short* const ring = reinterpret_cast<short*>(slds + 2 * a.lds_words);   // (RING only: launch_slim_shape sizes the dynamic LDS for it)
  const int lane = threadIdx.x;
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  const uint32_t w0 = PBH ? a.work_begin : 0u;
  constexpr int XREC = TWO ? BD::REC : BD::REC / 2;   // explicit history: ints per record (gap-affine: 4 x int16 per position; 2p: 16 bytes)
  const int max_records = PBH ? (int)min((long long)INT_MAX, a.pb_code_ints / (WI / 4))
                              : (XH ? (int)min((long long)INT_MAX, a.hist_stride / XREC) : INT_MAX);
  int* const xhist = XH ? a.hist + (long long)blockIdx.x * a.hist_stride : nullptr;   // explicit history: this workgroup's slice
  const int thr2 = 2 * a.max_dist_thr;
  const uint32_t wb = a.wbeg_dev ? *a.wbeg_dev : 0u;   // (the leftovers of one launch of the stage in front: BandArgs::wbeg_dev)
  for (uint32_t wi = w0 + wb + blockIdx.x; wi < w0 + nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    // (inl: the single-call form — meta, op-region offsets and the packed words of the one pair arrive in the kernel arguments)
    WfaPairMeta pm;
    int64_t coff_one[2] = {0, 0};
    if (inl) {   // (loads through a generic pointer: made wave-uniform by hand, or every mask below would live in vector registers)
      pm.p_woff = __builtin_amdgcn_readfirstlane(inl[0]); pm.t_woff = __builtin_amdgcn_readfirstlane(inl[1]);
      pm.plen = __builtin_amdgcn_readfirstlane(inl[2]); pm.tlen = __builtin_amdgcn_readfirstlane(inl[3]);
      coff_one[0] = ((int64_t)__builtin_amdgcn_readfirstlane(inl[5]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(inl[4]);
      coff_one[1] = ((int64_t)__builtin_amdgcn_readfirstlane(inl[7]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(inl[6]);
    } else {
      pm = a.meta[pair];
    }
    const int64_t* const coff = inl ? coff_one : a.cigar_off;
    const int plen = pm.plen, tlen = pm.tlen;
    const int ak = tlen - plen;
    uint8_t* const rec = PBH ? reinterpret_cast<uint8_t*>(a.hist + (long long)(wi - w0) * a.hist_stride) : nullptr;   // this pair's history slot
    const uint32_t* const wbase = inl ? inl + 8 : a.words;
    const uint32_t* gP = wbase + pm.p_woff;
    const uint32_t* gT = wbase + pm.t_woff;
    const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
    bool fallback = false;
    int pwb = 0, twb = 0;             // WIN: first staged word of each sequence
    int plim2 = INT_MAX, tlim2 = INT_MAX;   // WIN: the first (doubled) position whose probe would read past the staged words
    auto restage = [&](int np_, int nt_) {   // (uniform: every lane of the wave)
      pwb = np_; twb = nt_;
      __syncthreads();
      for (int i = lane; i < a.lds_words; i += 64) { const int gi = pwb + i; sP[i] = (gi < nwp) ? gP[gi] : 0u; }
      for (int i = lane; i < a.lds_words; i += 64) { const int gi = twb + i; sT[i] = (gi < nwt) ? gT[gi] : 0u; }
      __syncthreads();
      plim2 = (pwb + a.lds_words - 2) << 5; tlim2 = (twb + a.lds_words - 2) << 5;
    };
    if (max_records <= 1 || (TWO && 2 * max(plen, tlen) > 32000)) fallback = true;   // (2p: doubled offsets as int16)
    else if (WIN) restage(0, 0);
    else if (nwp + 3 > a.lds_words || nwt + 3 > a.lds_words) fallback = true;
    else {
      __syncthreads();
      for (int i = lane; i < nwp + 3; i += 64) sP[i] = (i < nwp) ? gP[i] : 0u;
      for (int i = lane; i < nwt + 3; i += 64) sT[i] = (i < nwt) ? gT[i] : 0u;
      __syncthreads();
    }
    if (RING) {   // every cell NULL (the pads stay so)
      for (int i = lane; i < RR * RW; i += 64) ring[i] = (short)-32768;
      __syncthreads();
    }
    int B = -(W / 2);  // diagonal of window position 0
    if (a.ef) {
      // wavefront 0 spans the diagonals [-pattern_begin_free, text_begin_free] (R/wavefront_aligner.c:259-302)
      if (a.pbf + a.tbf + 1 > W - 20) fallback = true;
      B = (a.tbf - a.pbf) / 2 - W / 2;
    }
    int result = 0, end_k = 0, end_off = 0, end_s = 0;
    int stop_status = 0, stop_score = 0;
    bool done = false;
    if (!fallback) {
      // per lane and chunk, doubled: kk2 = 2k; lim2 = 2 min(tlen, plen + k) (in-bounds <=> offset <= lim2; lim2c: not below 0);
      // dlim2 = 2 max(tlen, plen + k) (dlim2 - offset = distance to the end, R/wavefront_heuristic.c:176-192); ethr: the smallest
      // offset that ends the alignment on this diagonal
      int kk2[NCH], lim2[NCH], lim2c[NCH], dlim2[NCH], ethr[NCH];
      int cur[NCH], Mh[DM][NCH], Ih[E][NCH], Dh[E][NCH], I2h[E2D][NCH], D2h[E2D][NCH];
      int rw_off = 0;      // RING: cell offset of the row the next compute-next writes (row (s - x) mod RR)
      uint32_t hoff[NCH];  // piggy-back: byte of this diagonal in the history record compute-next fills: (step + 1) * WI + (k mod WI);
                           // explicit history: its position k mod WI in a record
      int step = 0;        // (the score of a step is step * g)
      // window geometry -> the per-lane constants
      auto set_lane_constants = [&]() {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int k = B + c * 64 + lane;
          kk2[c] = 2 * k; hoff[c] = (PBH ? (uint32_t)(step + 1) * WI : 0u) + ((uint32_t)k & (WI - 1));
          lim2[c] = 2 * min(tlen, plen + k); lim2c[c] = max(lim2[c], 0); dlim2[c] = 2 * max(tlen, plen + k);
          if (a.ef) {
            // (h >= tlen and plen - v <= pef) or (v >= plen and tlen - h <= tef), v = h - k (R/wavefront_termination.c:115-162):
            // two lower bounds on h each, the alignment ends at the smaller pair
            ethr[c] = 2 * min(max(tlen, plen - a.pef + k), max(plen + k, tlen - a.tef));
          } else {
            ethr[c] = (k == ak) ? 2 * tlen : INT_MAX;   // (R/wavefront_termination.c:37-61)
          }
        }
      };
      set_lane_constants();
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k = B + c * 64 + lane;
        cur[c] = (k == 0) ? 0 : NUL;  // wavefront 0
        if (a.ef && k >= -a.pbf && k <= a.tbf) cur[c] = 2 * max(k, 0);
#pragma unroll
        for (int j = 0; j < E; ++j) { Ih[j][c] = NUL; Dh[j][c] = NUL; }
#pragma unroll
        for (int j = 0; j < E2D; ++j) { I2h[j][c] = NUL; D2h[j][c] = NUL; }
#pragma unroll
        for (int j = 0; j < DM; ++j) Mh[j][c] = NUL;
      }
      // neighbour registers of the one-chunk form: lane 0 (from below) / lane 63 (from above) never receive a value
      int nb_mo_lo = NUL, nb_ie_lo = NUL, nb_mo_hi = NUL, nb_de_hi = NUL, nb_i2_lo = NUL, nb_d2_hi = NUL;
      int steps_wait = a.steps_between, dead_steps = 0;
      const int min_wf_len_m1 = a.min_wf_len - 1;
      // the first step the loop must not start: the step limit reached (score step * g >= max_steps) or no room for the record it fills
      const int step_stop = (int)min(min((long long)(max_records - 1), ((long long)a.max_steps + a.g - 1) / a.g), 1ll << 24);
      int akp = ak - B;           // window position of the end diagonal
      int act = NCH;              // active chunks: the hull of the ring (and 8 steps of growth either way) fits the first `act` chunks
      int leave = 0;              // 1 reached the end, 2 step limit, 3 hand the pair on
      int end_pos = 0, end_off2 = 0;   // leave == 1: window position and (doubled) offset of the cell that ended the alignment
#ifdef WFA_SLIM_COUNTERS
      uint32_t cnt_small = 0, cnt_big = 0, cnt_rounds = 0, cnt_cut = 0, cnt_shift = 0, cnt_oob = 0, cnt_live = 0;
#endif
      int togo = 0;               // steps left in the block of HP (an ending clears it)

      // One score step on ACT active chunks.  Every branch is a scalar branch and nothing leaves the step early: an ending sets
      // `leave` and the rest of the step is skipped, so the loops around it have one exit each.
      auto step_fn = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        // ---------------- extend M[s] (R/wavefront_extend_kernels.c:64-110) ----------------
        unsigned long long live[NCH], keep[NCH], hit[NCH];   // keep: lanes the cut-off keeps (all, unless it moves the wavefront's limits)
#pragma unroll
        for (int c = 0; c < NCH; ++c) { live[c] = 0ull; keep[c] = ~0ull; hit[c] = 0ull; }
        {
          // (a dead lane is parked at the end of its diagonal: nothing to compare, and its reads stay inside the staged words)
          int h2[NCH], v2[NCH];
#pragma unroll
          for (int c = 0; c < ACT; ++c) {
            h2[c] = (cur[c] >= 0) ? cur[c] : lim2c[c];
            v2[c] = h2[c] - kk2[c];
          }
#ifdef WFA_SLIM_COUNTERS
          if (ACT == 1) ++cnt_small; else ++cnt_big;
#endif
#ifndef WFA_SLIM_2P_SERIAL_EXT
#define WFA_SLIM_2P_SERIAL_EXT 0
#endif
          if (WFA_SLIM_2P_SERIAL_EXT && TWO && ACT > 1) {
            // (round 5, when gap-affine-2p kept 25 ring registers per chunk: the chunks extended one after the other, not interleaved, so
            // that one chunk's probe temporaries are live at a time; with the LDS ring of round 6 there is room to interleave them)
#pragma unroll
            for (int c = 0; c < ACT; ++c) {
              bool more;
              do {
#ifdef WFA_SLIM_COUNTERS
                ++cnt_rounds;
#endif
                const uint32_t m2 = min(slim_probe(sP, sT, v2[c], h2[c]), 64u);
                v2[c] += (int)m2; h2[c] += (int)m2;
                more = (m2 == 64u) && (h2[c] < lim2[c]);
              } while (__builtin_amdgcn_ballot_w64(more) != 0ull);
            }
          } else {
            bool more;
            bool go = true;   // (WIN: cleared when the live diagonals do not fit a window: the pair is handed on)
            do {
              more = false;
#ifdef WFA_SLIM_COUNTERS
              ++cnt_rounds;
#endif
              if (WIN) {
                auto past = [&]() {
                  bool o = false;
#pragma unroll
                  for (int c = 0; c < ACT; ++c) o |= (cur[c] >= 0) && (h2[c] >= tlim2 || v2[c] >= plim2);
                  return __builtin_amdgcn_ballot_w64(o) != 0ull;
                };
                if (past()) {   // (scalar branch) move both windows up to the smallest live positions
                  int mh = INT_MAX, mv = INT_MAX;
#pragma unroll
                  for (int c = 0; c < ACT; ++c) { const bool lv = cur[c] >= 0; mh = min(mh, lv ? h2[c] : INT_MAX); mv = min(mv, lv ? v2[c] : INT_MAX); }
                  mh = slim_wave_min(mh); mv = slim_wave_min(mv);
                  restage(max(0, (mv >> 5) - 1), max(0, (mh >> 5) - 1));
                  if (past()) { leave = 3; togo = 0; go = false; }
                }
              }
              if (go) {
#pragma unroll
                for (int c = 0; c < ACT; ++c) {
                  const uint32_t m2 = min(slim_probe(sP, sT, v2[c], h2[c], pwb, twb), 64u);
                  v2[c] += (int)m2; h2[c] += (int)m2;
                  more |= (m2 == 64u) && (h2[c] < lim2[c]);   // (past the end the zero padding of both sequences would match on)
                }
              }
            } while (go && __builtin_amdgcn_ballot_w64(more) != 0ull);
          }
#pragma unroll
          for (int c = 0; c < ACT; ++c) {
            int cc = cur[c];
            asm("" : "+v"(cc));   // (a compare of its own: carried across the loop above, the first one's mask takes a round trip through a VGPR)
            const bool lv = cc >= 0;
            live[c] = __builtin_amdgcn_ballot_w64(lv);
            cur[c] = lv ? min(h2[c], lim2[c]) : cc;
          }
        }
        unsigned long long any_live = 0ull;
#pragma unroll
        for (int c = 0; c < ACT; ++c) any_live |= live[c];
        if (WIN && leave != 0) any_live = 0ull;   // (handed on inside the extension: nothing of this step counts)
        if (any_live) {
          dead_steps = 0;
#ifdef WFA_SLIM_COUNTERS
#pragma unroll
          for (int c = 0; c < ACT; ++c) cnt_live += (uint32_t)__builtin_popcountll(live[c]);
#endif
          // ---------------- termination: the lowest diagonal whose offset reaches its threshold ----------------
          unsigned long long any_hit = 0ull;
#pragma unroll
          for (int c = 0; c < ACT; ++c) { hit[c] = __builtin_amdgcn_ballot_w64(cur[c] >= ethr[c]); any_hit |= hit[c]; }
          --steps_wait;
          if (any_hit) {
            leave = 1; togo = 0; end_pos = BD::first_pos(hit);
#pragma unroll
            for (int c = 0; c < ACT; ++c) if ((end_pos >> 6) == c) end_off2 = __builtin_amdgcn_readlane(cur[c], end_pos & 63);
          }
          // ---------------- wf-adaptive cut-off (R/wavefront_heuristic.c:257-293, 509-567); a.heur == 0: no heuristic ----------------
          else if (a.heur == 1 && steps_wait <= 0) {
            const int lo_p = BD::first_pos(live), hi_p = BD::last_pos(live);   // window positions of the wavefront's ends
            if (hi_p - lo_p >= min_wf_len_m1) {
              int d[NCH], dm = 0x7fffffff;
#pragma unroll
              for (int c = 0; c < ACT; ++c) { d[c] = dlim2[c] - cur[c]; dm = min(dm, d[c]); }   // 2 max(plen - v, tlen - h); dead lanes ~ 2^30
              const int dmin = slim_wave_min(dm);   // (a live lane's distance never exceeds max(plen, tlen): the reference's initial value cannot win)
              unsigned long long ok[NCH];
#pragma unroll
              for (int c = 0; c < NCH; ++c) ok[c] = 0ull;
#pragma unroll
              for (int c = 0; c < ACT; ++c) ok[c] = __builtin_amdgcn_ballot_w64(d[c] <= dmin + thr2);
              // (the lane of the minimum always qualifies: ok is never empty)
              int fp, lp;
              if (ACT == 1) { fp = (int)__builtin_ctzll(ok[0]); lp = 63 - (int)__builtin_clzll(ok[0]); }
              else { fp = BD::first_pos(ok); lp = BD::last_pos(ok); }
              int new_lo = lo_p, new_hi = hi_p;
              const int top_limit = min(akp, hi_p);
              if (top_limit > lo_p) new_lo = min(fp, top_limit);
              const int bottom_limit = max(akp, new_lo);
              if (bottom_limit < hi_p) new_hi = max(lp, bottom_limit);
              steps_wait = a.steps_between;
              // the wavefront's limits changed <=> a live lane lies outside them; then M and the gap components are cut to them (the equate)
              if (ACT == 1) {
                const unsigned long long kp = (~0ull << new_lo) & (~0ull >> (63 - new_hi));
                if (live[0] & ~kp) keep[0] = kp;
#ifdef WFA_SLIM_COUNTERS
                if (live[0] & ~kp) ++cnt_cut;
#endif
              } else if (new_lo != lo_p || new_hi != hi_p) {
#pragma unroll
                for (int c = 0; c < ACT; ++c) keep[c] = slim_range_mask(c, new_lo, new_hi);
              }
            }
          }
        } else {
          // nothing alive at this score: the first scores of the lattice; a ring that stays dead is left to the next stage
          if (++dead_steps > 2 * DM + 2 + (TWO ? OE2 : 0)) { leave = 3; togo = 0; }
        }
        if (leave == 0) {
          // (the cut, in each value's own register and without a branch: the paths of the step join on scalars only)
#pragma unroll
          for (int c = 0; c < ACT; ++c) {
            slim_keep(cur[c], keep[c]); slim_keep(Ih[0][c], keep[c]); slim_keep(Dh[0][c], keep[c]);
            if (TWO) { slim_keep(I2h[0][c], keep[c]); slim_keep(D2h[0][c], keep[c]); }
          }
          if (XH) {
            // explicit history of score s (after the cut-off, so dropped lanes read NULL): one entry per window position, the layout
            // band_backtrace reads (wfa_band.hpp: int16 halves, negative -> -1; offsets are doubled here); inactive chunks hold nothing
            int* const xr = xhist + (long long)step * XREC;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const bool on = c < ACT;
              const int m_ = on ? sat16(cur[c] >> 1) : -1, i_ = on ? sat16(Ih[0][c] >> 1) : -1, d_ = on ? sat16(Dh[0][c] >> 1) : -1;
              if (TWO) {
                const int i2_ = on ? sat16(I2h[0][c] >> 1) : -1, d2_ = on ? sat16(D2h[0][c] >> 1) : -1;
                reinterpret_cast<int4*>(xr)[hoff[c]] = make_int4((m_ & 0xffff) | (i_ << 16), (d_ & 0xffff) | (B << 16), (i2_ & 0xffff) | (d2_ << 16), 0);
              } else {
                reinterpret_cast<short4*>(xr)[hoff[c]] = make_short4((short)m_, (short)i_, (short)d_, (short)B);
              }
            }
          }
          // ---------------- compute-next for score s + g (R/wavefront_compute_affine.c:44-86, R/wavefront_compute_affine2p.c:45-106) ----------------
          // RING: the value leaving the registers, M[s - x], goes to the row M[s - o2 - e2] held (written before this step's reads: with
          // o + e = x + 1 the shallow read is that very row); the reads are rows (s + 1 - o - e) and (s + 1 - o2 - e2)
          int roe_off = 0, roe2_off = 0;
          if (RING) {
            short* const wrow = ring + rw_off + 1 + lane;
#pragma unroll
            for (int c = 0; c < ACT; ++c) wrow[64 * c] = (short)max(Mh[DM - 1][c], -32768);
            roe_off = rw_off - (OE - 1 - X) * RW; roe_off += (roe_off < 0) ? RR * RW : 0;
            roe2_off = rw_off + RW; roe2_off = (roe2_off == RR * RW) ? 0 : roe2_off;   // (row s + 1 - o2 - e2 = the next row to be written)
            rw_off = roe2_off;
          }
#pragma unroll
          for (int j = DM - 1; j > 0; --j)
#pragma unroll
            for (int c = 0; c < ACT; ++c) Mh[j][c] = Mh[j - 1][c];
#pragma unroll
          for (int c = 0; c < ACT; ++c) Mh[0][c] = cur[c];
          int ni[NCH], nd[NCH], ni2[NCH], nd2[NCH];
          unsigned long long oob = 0;
          constexpr int MOD = RING ? 0 : OE - 1;   // (register depth of M[s + 1 - o - e]; RING reads it from LDS)
#pragma unroll
          for (int c = 0; c < ACT; ++c) {
            int mo_lo, ie_lo, mo_hi, de_hi, i2e_lo = NUL, d2e_hi = NUL;
            int mo2_lo = NUL, mo2_hi = NUL;
            if (RING) {
              // cell of window position p = 64 c + lane is at index p + 1: its neighbours k - 1 / k + 1 at p and p + 2 (pads NULL)
              const short* const r1 = ring + roe_off + 64 * c + lane;
              const short* const r2 = ring + roe2_off + 64 * c + lane;
              mo_lo = r1[0]; mo_hi = r1[2]; mo2_lo = r2[0]; mo2_hi = r2[2];
            }
            if (ACT == 1) {
              if (!RING) {
                nb_mo_lo = __builtin_amdgcn_update_dpp(nb_mo_lo, Mh[MOD][0], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                nb_mo_hi = __builtin_amdgcn_update_dpp(nb_mo_hi, Mh[MOD][0], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                mo_lo = nb_mo_lo; mo_hi = nb_mo_hi;
              }
              nb_ie_lo = __builtin_amdgcn_update_dpp(nb_ie_lo, Ih[E - 1][0], 0x138, 0xf, 0xf, false);
              nb_de_hi = __builtin_amdgcn_update_dpp(nb_de_hi, Dh[E - 1][0], 0x130, 0xf, 0xf, false);
              ie_lo = nb_ie_lo; de_hi = nb_de_hi;
              if (TWO) {
                nb_i2_lo = __builtin_amdgcn_update_dpp(nb_i2_lo, I2h[E2D - 1][0], 0x138, 0xf, 0xf, false);
                nb_d2_hi = __builtin_amdgcn_update_dpp(nb_d2_hi, D2h[E2D - 1][0], 0x130, 0xf, 0xf, false);
                i2e_lo = nb_i2_lo; d2e_hi = nb_d2_hi;
              }
            } else {
              if (!RING) { mo_lo = BD::below(Mh[MOD], c); mo_hi = BD::above(Mh[MOD], c); }
              ie_lo = BD::below(Ih[E - 1], c);
              de_hi = BD::above(Dh[E - 1], c);
              if (TWO) { i2e_lo = BD::below(I2h[E2D - 1], c); d2e_hi = BD::above(D2h[E2D - 1], c); }
            }
            ni[c] = max(mo_lo, ie_lo) + 2;
            nd[c] = max(mo_hi, de_hi);
            const int x1 = Mh[X - 1][c] + 2;
            int t = max(ni[c], nd[c]);
            ni2[c] = NUL; nd2[c] = NUL;
            if (TWO) {
              ni2[c] = max(mo2_lo, i2e_lo) + 2;
              nd2[c] = max(mo2_hi, d2e_hi);
              t = max(t, max(ni2[c], nd2[c]));
            }
            const int m = max(x1, t);
            if (PBH) {
              // the choice the backtrace would make (R/wavefront_backtrace.c:49-59: mismatch > D2 > D1 > I2 > I1, extension > opening on
              // equal offsets), taken where the candidates are in registers — the piggy-back history of score s + g: one byte per
              // active diagonal, stored here (the window may move before the next step begins; the record of score 0 is never read).
              // BandArgs::pb_raw: the sign bits of the subtractions, shifted in one after the other
              uint32_t cd;
              if (TWO) {
                cd = (uint32_t)(x1 - m) >> 31;                                              // bit 7: the mismatch is below the best
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(nd2[c] - m), 31);              // 6: D2 below the best
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(nd[c] - m), 31);               // 5: D1 below the best
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(ni2[c] - m), 31);              // 4: I2 below the best (then I1 made it)
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(ie_lo - mo_lo), 31);           // 3: I1 opened (its extension is below)
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(de_hi - mo_hi), 31);           // 2: D1 opened
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(i2e_lo - mo2_lo), 31);         // 1: I2 opened
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(d2e_hi - mo2_hi), 31);         // 0: D2 opened
              } else {
                cd = (uint32_t)(x1 - t) >> 31;                                               // bit 3: the mismatch is below the best gap
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(nd[c] - ni[c]), 31);            // 2: the deletion is below the insertion
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(ie_lo - mo_lo), 31);            // 1: I opened
                cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(de_hi - mo_hi), 31);            // 0: D opened
              }
              rec[hoff[c]] = (uint8_t)cd;
              hoff[c] += WI;   // (an inactive chunk's offset is set again when it joins)
            }
            // only M is clamped.  RING: a dead cell may be a small negative number here (NULL is read from the ring as -32768), and M must
            // stay exact (its values go back into the ring, the hull and the cut-off): one unsigned compare covers both ends (lim2c =
            // max(lim2, 0); a diagonal with lim2 < 0 holds no cell, and nothing can reach 0 there)
            if (RING) cur[c] = ((uint32_t)m > (uint32_t)lim2c[c]) ? NUL : m;
            else cur[c] = (m > lim2[c]) ? NUL : m;  // (negative values are dead already)
            oob |= __builtin_amdgcn_ballot_w64(t > lim2[c]);
          }
#pragma unroll
          for (int j = E - 1; j > 0; --j)
#pragma unroll
            for (int c = 0; c < ACT; ++c) { Ih[j][c] = Ih[j - 1][c]; Dh[j][c] = Dh[j - 1][c]; }
#pragma unroll
          for (int c = 0; c < ACT; ++c) { Ih[0][c] = ni[c]; Dh[0][c] = nd[c]; }
          if (TWO) {
#pragma unroll
            for (int j = E2D - 1; j > 0; --j)
#pragma unroll
              for (int c = 0; c < ACT; ++c) { I2h[j][c] = I2h[j - 1][c]; D2h[j][c] = D2h[j - 1][c]; }
#pragma unroll
            for (int c = 0; c < ACT; ++c) { I2h[0][c] = ni2[c]; D2h[0][c] = nd2[c]; }
          }
#ifdef WFA_SLIM_COUNTERS
          if (oob) ++cnt_oob;
#endif
          if (oob) {
            // trim the ends of the gap components (R/wavefront_compute.c:571-605): outside [first, last] in-bounds -> NULL
            auto trim = [&](int (&g)[NCH]) {
              unsigned long long b[NCH];
#pragma unroll
              for (int c = 0; c < NCH; ++c) b[c] = 0ull;
#pragma unroll
              for (int c = 0; c < ACT; ++c) b[c] = __builtin_amdgcn_ballot_w64(g[c] >= 0 && g[c] <= lim2[c]);
              const int lo = BD::first_pos(b), hi = BD::last_pos(b);
#pragma unroll
              for (int c = 0; c < ACT; ++c) {
                const int p = c * 64 + lane;
                g[c] = (p < lo || p > hi) ? NUL : g[c];
              }
            };
            trim(Ih[0]); trim(Dh[0]);
            if (TWO) { trim(I2h[0]); trim(D2h[0]); }
          }
          ++step;
          // ---------------- step limit (R/wavefront_unialign.c:98-107), room for the next step's history record ----------------
          if (step >= step_stop) { leave = ((long long)step * a.g >= (long long)a.max_steps) ? 2 : 3; togo = 0; }
        }
      };

      // every register of the chunks from `c0` on: NULL
      auto null_chunks = [&](int c0) {
#pragma unroll
        for (int c = 1; c < NCH; ++c) if (c >= c0) {
          cur[c] = NUL;
#pragma unroll
          for (int j = 0; j < E; ++j) { Ih[j][c] = NUL; Dh[j][c] = NUL; }
#pragma unroll
          for (int j = 0; j < E2D; ++j) { I2h[j][c] = NUL; D2h[j][c] = NUL; }
#pragma unroll
          for (int j = 0; j < DM; ++j) Mh[j][c] = NUL;
          // (RING: the cells of an inactive chunk are NULL already — only active chunks are written, and what a shift or a smaller
          // form leaves beyond them lay outside the hull)
        }
      };
      while (leave == 0) {
        // ---------------- keep the ring inside the window: every HP steps (growth is <= 1 diagonal per step and side) ----------------
        {
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
            for (int j = 0; j < DM - 1; ++j) any &= Mh[j][c];   // (gap-affine: the oldest M is dropped by the next compute-next)
            if (TWO) {
              any &= Mh[DM - 1][c];
              // the rows of the ring (an inactive chunk's cells are NULL: not read); and the gap rings are cleaned: a dead cell computed from
              // a NULL of the ring is a small negative number that would otherwise creep upwards by 2 per step
              if (c < act) {
#pragma unroll 3
                for (int r = 0; r < RR; ++r) any &= (int)ring[r * RW + 1 + 64 * c + lane];
              }
#pragma unroll
              for (int j = 0; j < E; ++j) { Ih[j][c] = (Ih[j][c] < 0) ? NUL : Ih[j][c]; Dh[j][c] = (Dh[j][c] < 0) ? NUL : Dh[j][c]; }
#pragma unroll
              for (int j = 0; j < E2D; ++j) { I2h[j][c] = (I2h[j][c] < 0) ? NUL : I2h[j][c]; D2h[j][c] = (D2h[j][c] < 0) ? NUL : D2h[j][c]; }
            }
            hull[c] = (c < act) ? __builtin_amdgcn_ballot_w64(any >= 0) : 0ull;  // (an inactive chunk's registers are stale, not read)
          }
          const int fp = BD::first_pos(hull), lp = BD::last_pos(hull);
          if (lp >= 0) {
            const int width = lp - fp + 1;
            if (width > W - 20) { leave = 3; }
            // the fewest chunks that hold the hull and HP steps of growth either way; a little hysteresis keeps a hull near a limit from
            // being shifted to and fro
            int want = NCH;
#pragma unroll
            for (int n = NCH - 1; n >= 1; --n) if (width <= 64 * n - 2 * HP - (n < act ? 6 : 2)) want = n;
            const int wa = 64 * want;
            const bool shift = fp < HP + 1 || lp > wa - HP - 2;
            if (leave == 0 && (shift || want != act)) {
              if (!(shift && act == 1 && want == 1)) null_chunks(act);   // (the chunks beyond the active ones hold stale values: they read as NULL)
              if (shift) {
#ifdef WFA_SLIM_COUNTERS
                ++cnt_shift;
#endif
                const int delta = fp - (wa - width) / 2;   // re-centre in the chunks of the form that follows
                B += delta; akp -= delta;
                if (act == 1 && want == 1) {
                  // 64 diagonals before and after: one lane shuffle per register
                  const int src = lane + delta;
                  const bool in = (unsigned)src < 64u;
                  auto sh1 = [&](int& r, int nullv) { const int t = __shfl(r, src & 63, 64); r = in ? t : nullv; };
                  sh1(cur[0], NUL);
#pragma unroll
                  for (int j = 0; j < E; ++j) { sh1(Ih[j][0], NUL); sh1(Dh[j][0], NUL); }
                  if (TWO) {
#pragma unroll
                    for (int j = 0; j < E2D; ++j) { sh1(I2h[j][0], NUL); sh1(D2h[j][0], NUL); }
                  }
#pragma unroll
                  for (int j = 0; j < DM - 1; ++j) sh1(Mh[j][0], NUL);
                  if (TWO) {
                    sh1(Mh[DM - 1][0], NUL);
                    // the ring rows: window position p takes what p + delta held (one chunk before and after: the source is inside it or NULL)
#pragma unroll 1
                    for (int r = 0; r < RR; ++r) {
                      short* const row = ring + r * RW + 1;
                      const short t = in ? row[src & 63] : (short)-32768;
                      row[lane] = t;
                    }
                  }
                } else {
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
#pragma unroll 1
                    for (int r = 0; r < RR; ++r) {
                      short* const row = ring + r * RW + 1;
                      short t[NCH];
#pragma unroll
                      for (int c = 0; c < NCH; ++c) { const int sp = 64 * c + lane + delta; t[c] = ((unsigned)sp < (unsigned)W) ? row[sp] : (short)-32768; }
#pragma unroll
                      for (int c = 0; c < NCH; ++c) row[64 * c + lane] = t[c];
                    }
                  }
                }
              }
              set_lane_constants();
              act = want;
            }
          }
        }
        togo = (leave == 0) ? HP : 0;
        if (act == 1) {
#pragma unroll 1
          while (togo > 0) { --togo; step_fn(band_int<1>{}); }
        } else if (NCH == 2 || act == 2) {
#pragma unroll 1
          while (togo > 0) { --togo; step_fn(band_int<2>{}); }
        } else if (NCH == 3 || act == 3) {
#pragma unroll 1
          while (togo > 0) { --togo; step_fn(band_int<(NCH > 3 ? 3 : NCH)>{}); }
        } else {
#pragma unroll 1
          while (togo > 0) { --togo; step_fn(band_int<NCH>{}); }
        }
      }
#ifdef WFA_SLIM_COUNTERS
      if (a.dbg && lane == 0) {
        atomicAdd(a.dbg + 0, cnt_small); atomicAdd(a.dbg + 1, cnt_big); atomicAdd(a.dbg + 2, cnt_rounds); atomicAdd(a.dbg + 3, cnt_cut);
        atomicAdd(a.dbg + 4, cnt_shift); atomicAdd(a.dbg + 5, cnt_oob); atomicAdd(a.dbg + 6, cnt_live >> 6);
      }
#endif
      const int s_end = step * a.g;
      if (leave == 1) {
        done = true; result = -s_end; end_s = s_end; end_k = B + end_pos; end_off = end_off2 >> 1;
      }
      else if (leave == 2) { stop_status = WFA_STATUS_MAX_STEPS_REACHED; stop_score = -a.max_steps; }
      else fallback = true;
    }
    if (FULL && stop_status != 0 && lane == 0) {   // no end cell: no walk, empty op string (R/wavefront_unialign.c:147-237)
      a.cigar_begin[pair] = coff[pair + 1];
      a.cigar_len[pair] = 0;
    }
    if (PBH && lane == 0) a.end_state[wi - w0] = make_int4(end_s, end_k, end_off, (fallback || stop_status != 0 || !done) ? 0 : 1);
    if (XH && done && !fallback && stop_status == 0) {
      __syncthreads();   // this wave's history stores before its own loads
      long long begin = 0;
      uint8_t* buf = a.cigar_ops + coff[pair];
      band_backtrace<NCH>(xhist, a, plen, tlen, end_s, end_k, end_off, buf, &begin, lane, 64);
      if (lane == 0) {
        a.cigar_begin[pair] = coff[pair] + begin;
        a.cigar_len[pair] = (int)((long long)plen + tlen - begin);
      }
    }
    if (a.done) __threadfence_system();   // the single-call path: the op bytes of every lane before the flag below
    if (lane == 0) {
      if (res2) { res2[0] = fallback ? 0 : (stop_status != 0 ? stop_score : result); res2[1] = fallback ? WFA_INTERNAL_FALLBACK : stop_status; }
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
      // (the host polls this flag in the pinned block instead of waiting for the stream)
      if (a.done) { __threadfence_system(); __hip_atomic_store(&a.done[pair], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
}

// waves per SIMD of the 128-diagonal form by the registers a shape's rings take (two chunks of max(x, o+e) + 2 e each): the library's
// shapes (<= 18) run seven; deeper rings compiled at run time get the registers they need instead of spilling
constexpr int slim_waves(int X, int OE, int E) {
  const int ring = 2 * ((X > OE ? X : OE) + 2 * E);
  return ring <= 18 ? WFA_SLIM_WAVES : ring <= 24 ? 6 : ring <= 32 ? 5 : 4;
}
template <int NCH, int HIST, int X, int OE, int E, int OE2, int E2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(slim_waves(X, OE, E), slim_waves(X, OE, E))))
wfa_slim_kernel(const BandArgs a) {   // gap-affine, 128 diagonals: seven waves per SIMD (<= 72 VGPRs; the sequences of 10 kb reads in LDS allow eight)
  wfa_slim_body<NCH, HIST, X, OE, E, OE2, E2>(a);
}
template <int NCH, int HIST, int X, int OE, int E, int OE2, int E2, bool WIN = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
wfa_slim_kernel_tail(const BandArgs a) {   // 256 diagonals, the stage behind the first window: few pairs; at least two waves per SIMD
  wfa_slim_body<NCH, HIST, X, OE, E, OE2, E2, WIN>(a);   // (gap-affine-2p would take 297 registers: C4's tail is a few thousand pairs)
}
template <int NCH, int HIST, int X, int OE, int E, int OE2, int E2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WFA_SLIM_WAVES_2P, WFA_SLIM_WAVES_2P)))
wfa_slim_kernel_2p(const BandArgs a) {   // gap-affine-2p: 25 ring registers per chunk; three waves per SIMD (<= 168 VGPRs) measured best (C4 with wf-adaptive, 10 k pairs: 21.3 ms against 25.8 ms at four waves with spills and 23.2 ms at two)
  wfa_slim_body<NCH, HIST, X, OE, E, OE2, E2>(a);
}

// One pair per call (wfa_hip_align_pair): everything the wave reads from the host — lengths, op-region offsets, the packed words —
// rides in the kernel arguments, so the kernel starts without a load from the pinned block (two dependent PCIe round trips less).
// (SlimOne, the block of one pair: wfa_band.hpp)
template <int NCH, int HIST, int X, int OE, int E, int OE2, int E2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WFA_SLIM_WAVES, WFA_SLIM_WAVES)))
wfa_slim_kernel_one(const BandArgs a, const SlimOne q) {
  typedef const char __attribute__((address_space(4))) * kernarg_ptr;
  constexpr size_t off = (sizeof(BandArgs) + alignof(SlimOne) - 1) & ~(alignof(SlimOne) - 1);
  const uint32_t* inl = (const uint32_t*)((kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + off);
  wfa_slim_body<NCH, HIST, X, OE, E, OE2, E2>(a, inl);
}

// The same, resident (round 6): pywfa's own usage is one wavefront_align(text) per call, 1-2 us on a host core; a
// kernel launch per call costs ~13 us launch-to-flag on this platform.  A one-wave instance of this kernel stays on the device between
// calls and takes its pairs from a MAILBOX in pinned host memory: the host writes the pair (the block wfa_slim_kernel_one gets as kernel
// arguments) and then the request number; the wave polls the number at system scope (s_sleep between reads), aligns, writes results and
// op bytes into the pinned block as the one-pair kernel does, and publishes the request number as done.  It leaves by itself after
// `idle_ticks` of the 100 MHz clock without a request (or when told to quit): nothing of it outlives an idle aligner or a process that
// forgets to close one, and the host starts a new instance with the next call.
// (SlimMailbox: wfa_band.hpp)
template <int NCH, int HIST, int X, int OE, int E, int OE2, int E2>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WFA_SLIM_WAVES, WFA_SLIM_WAVES)))
wfa_slim_kernel_mailbox(const BandArgs a, SlimMailbox* const mb) {
  extern __shared__ uint32_t slds[];
  uint32_t* const rq = slds + 2 * a.lds_words;   // the request, de-tagged: the block wfa_slim_body takes (launch_slim_mailbox_shape sizes the LDS for it)
  const int lane = threadIdx.x;
  uint32_t last = __builtin_amdgcn_readfirstlane((uint32_t)__hip_atomic_load(&mb->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) & 0xFFFFFFu;
  const uint64_t idle = mb->idle_ticks;
  uint32_t served = 0;
  // the request: WFA_MB_LINES lines of 15 data words + the request number in the 16th.  A line is read (and written) as a whole, so a
  // line that carries the awaited number carries that request's words: ONE round trip over PCIe both finds the request and fetches it
  const uint32_t* const rl = &mb->req[0][0];
  const bool t0 = (lane & 15) == 15;   // this lane's word of a 64-word slice is a line's number
  for (;;) {
    const uint64_t tbeg = wall_clock64();
    const uint32_t want = (last + 1u) & 0xFFFFFFu;
    uint64_t polls = 0;   // (a bound of its own on the wait, should the clock not be what it is taken for: a poll is a PCIe round trip, > 0.25 us)
    uint32_t v0, v1, v2;
    for (;;) {
      v0 = __hip_atomic_load(rl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v1 = __hip_atomic_load(rl + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v2 = (lane < 16 * WFA_MB_LINES - 128) ? __hip_atomic_load(rl + 128 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : want;
      const bool stale = t0 && (v0 != want || v1 != want || v2 != want);
      if (__builtin_amdgcn_ballot_w64(stale) == 0ull) break;
      const uint32_t q = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&mb->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
      if (q != 0 || wall_clock64() - tbeg > idle || ++polls > idle / 16 + 4096) {
        // (a request posted after the reads above is not lost: the host sees alive == 0 with its request not done and starts an instance)
        if (lane == 0) __hip_atomic_store(&mb->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    const uint64_t t_req = wall_clock64();
    __syncthreads();   // (the previous request's readers of rq are done)
    if (!t0) { rq[(lane >> 4) * 15 + (lane & 15)] = v0; rq[((64 + lane) >> 4) * 15 + (lane & 15)] = v1; }
    if (!t0 && lane < 16 * WFA_MB_LINES - 128) rq[((128 + lane) >> 4) * 15 + (lane & 15)] = v2;
    __syncthreads();
    int res2[2] = {0, 0};
    wfa_slim_body<NCH, HIST, X, OE, E, OE2, E2>(a, rq, res2);
    if (HIST != 0) __threadfence_system();   // op bytes, their begin and length, of every lane before the answer below
    ++served;
    if (lane == 0) {
      // the answer — request number, status code, score — is ONE 8-byte store: score-only needs nothing else, and no fence before it
      const uint32_t code = (res2[1] == 0) ? 0u : (res2[1] == WFA_INTERNAL_FALLBACK) ? 255u : 2u;   // (the kernel's statuses: done, handed on, step limit)
      const unsigned long long ans = ((unsigned long long)(uint32_t)res2[0] << 32) | (code << 24) | want;
      mb->served = served;
      mb->ticks = (uint32_t)(wall_clock64() - t_req);   // (diagnostics: 10 ns ticks from the request's arrival to its answer)
      if (HIST != 0) __hip_atomic_store(&mb->done, ans, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      else __hip_atomic_store(&mb->done, ans, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    last = want;
  }
}

#ifndef __HIPCC_RTC__
// starts an instance of the resident kernel (gap-affine shapes of the library, 128 diagonals; score-only or the explicit history walked
// in-kernel); -1: this launch has no such form (the caller takes the launch-per-call path)
template <int X, int OE, int E, int OE2, int E2>
static int launch_slim_mailbox_shape(const BandArgs& a, bool full, hipStream_t stream, SlimMailbox* mb) {
  if constexpr (OE2 > 0) { return -1; }
  else {
    const size_t smem = (size_t)a.lds_words * 2 * sizeof(uint32_t) + (size_t)(15 * WFA_MB_LINES + 16) * sizeof(uint32_t);   // + the de-tagged request
    if (full) hipLaunchKernelGGL((wfa_slim_kernel_mailbox<2, 2, X, OE, E, 0, 0>), dim3(1), dim3(64), smem, stream, a, mb);
    else hipLaunchKernelGGL((wfa_slim_kernel_mailbox<2, 0, X, OE, E, 0, 0>), dim3(1), dim3(64), smem, stream, a, mb);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
}

template <int X, int OE, int E, int OE2, int E2>
static int launch_slim_shape(const BandArgs& a, int nch, bool full, long long grid, hipStream_t stream) {
  constexpr int NCH1 = (OE2 > 0) ? 3 : 2;   // the first window: 128 diagonals, gap-affine-2p 192
  size_t smem = (size_t)a.lds_words * 2 * sizeof(uint32_t);
  if (OE2 > 0) smem += (size_t)(OE2 - X) * (size_t)(64 * (nch == 4 ? 4 : NCH1) + 2) * sizeof(short);   // gap-affine-2p: the LDS ring of the deep M history
  static const int pad_kb = getenv("WFA_HIP_SLIM_LDS_PAD_KB") ? atoi(getenv("WFA_HIP_SLIM_LDS_PAD_KB")) : 0;   // (occupancy experiments)
  smem += (size_t)pad_kb << 10;
  if (nch == 4) {
    // the stage behind it: 256 diagonals, explicit history walked in-kernel
    // (round 5: also as the FIRST stage of reads over 20 kb with CIGARs — piggy-back history of a split launch)
    if constexpr (OE2 == 0) {
      // reads beyond what LDS holds: the windowed form (first stage of reads over 26 kb: score only, or the piggy-back history)
      if (a.win && (!full || a.split)) {
        if (full) hipLaunchKernelGGL((wfa_slim_kernel_tail<4, 1, X, OE, E, OE2, E2, true>), dim3((unsigned)grid), dim3(64), smem, stream, a);
        else hipLaunchKernelGGL((wfa_slim_kernel_tail<4, 0, X, OE, E, OE2, E2, true>), dim3((unsigned)grid), dim3(64), smem, stream, a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
      }
    }
    if (full && a.split) hipLaunchKernelGGL((wfa_slim_kernel_tail<4, 1, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else if (full) hipLaunchKernelGGL((wfa_slim_kernel_tail<4, 2, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else hipLaunchKernelGGL((wfa_slim_kernel_tail<4, 0, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
  } else if constexpr (OE2 > 0) {
    if (full && a.split) hipLaunchKernelGGL((wfa_slim_kernel_2p<NCH1, 1, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else if (full) hipLaunchKernelGGL((wfa_slim_kernel_2p<NCH1, 2, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else hipLaunchKernelGGL((wfa_slim_kernel_2p<NCH1, 0, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
  } else if (a.one && !a.split && grid == 1) {
    const SlimOne& q = *reinterpret_cast<const SlimOne*>(a.one);
    if (full) hipLaunchKernelGGL((wfa_slim_kernel_one<NCH1, 2, X, OE, E, OE2, E2>), dim3(1), dim3(64), smem, stream, a, q);
    else hipLaunchKernelGGL((wfa_slim_kernel_one<NCH1, 0, X, OE, E, OE2, E2>), dim3(1), dim3(64), smem, stream, a, q);
  } else {
    // (unsplit with a history: reads of up to 1 kb, the single-call path — explicit history, walked in-kernel)
    if (full && a.split) hipLaunchKernelGGL((wfa_slim_kernel<NCH1, 1, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else if (full) hipLaunchKernelGGL((wfa_slim_kernel<NCH1, 2, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
    else hipLaunchKernelGGL((wfa_slim_kernel<NCH1, 0, X, OE, E, OE2, E2>), dim3((unsigned)grid), dim3(64), smem, stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

}  // namespace wfa
