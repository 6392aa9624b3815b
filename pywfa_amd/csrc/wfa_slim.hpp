// wfa_slim.hpp — the slim form of the banded kernel (wfa_band.hpp): the step of the long-read hot path (BASELINE C3: 10 kb reads,
// gap-affine, wf-adaptive, full CIGAR) rebuilt around its instruction count.  Same algorithm, same window (128 diagonals that
// slide, 64 of them active while the hull of the ring fits), same piggy-back history and the same results as
// wfa_band_kernel<2, FULL, true, true, FULL, FULL, X, OE, E, 0, 0>; what differs is what a wave issues per score step
// (R = /root/reference/pywfa/WFA2_lib/wavefront):
//   * offsets are kept DOUBLED (2 x h): an offset is then the bit position of its base in the 2-bit packed text, the extension
//     (R/wavefront_extend_kernels.c:64-110) needs no conversion on the way in or out, and every comparison of the step is
//     invariant under the scaling;
//   * no divergent branch inside the step: every branch of the loop is a scalar branch, so the compiler keeps the control flow as
//     written (one lane-dependent `if` makes it linearise the whole loop body behind flag registers);
//   * the wf-adaptive cut-off (R/wavefront_heuristic.c:176-293) takes its wave minimum in six DPP steps, its limits in window
//     positions from two 64-bit masks, and drops lanes through a mask — no per-chunk position arithmetic in the 64-diagonal form;
//   * the k-1 / k+1 neighbours (R/wavefront_compute_affine.c:44-86) come through registers whose edge lane is NULL for good, so a
//     shift is one DPP move;
//   * the 64- and the 128-diagonal form are two straight step bodies, chosen once per block of 8 steps at the hull check, so the
//     inactive chunk costs nothing;
//   * one piggy-back byte per ACTIVE diagonal and step is stored (64 B instead of 128 B per step in the small form).
// Scope: gap-affine, match = 0, wf-adaptive, end-to-end, sequences staged in LDS (reads <= 10 kb), score-only or piggy-back
// history of a split launch.  Everything else stays with wfa_band_kernel; a pair whose window overflows is handed on exactly as
// there.
#pragma once
#include "wfa_band.hpp"

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

// one probe of the extension on doubled coordinates: v2 / h2 = bit positions in the packed pattern / text (LDS byte offsets
// sp / st); returns the number of matching BITS (even, <= 64) before the first difference
__device__ __forceinline__ uint32_t slim_probe(const uint32_t* sP, const uint32_t* sT, int v2, int h2) {
  const int pi = v2 >> 5, ti = h2 >> 5;
  const uint32_t p0 = sP[pi], p1 = sP[pi + 1], p2 = sP[pi + 2], t0 = sT[ti], t1 = sT[ti + 1], t2 = sT[ti + 2];
  const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v2) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h2);
  const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v2) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h2);
  return min(band_ffbl(xl), band_ffbl(xh) | 32u) & ~1u;   // (v_ffbl_b32 gives ~0 for 0)
}

template <bool FULL, int X, int OE, int E>
__device__ __forceinline__ void wfa_slim_body(const BandArgs& a) {
  constexpr int NCH = 2;
  typedef Band<NCH> BD;
  constexpr int W = 128, WI = 128, WS = 64, HP = 8;
  constexpr int DM = (X > OE) ? X : OE;
  constexpr int NUL = WFA_OFFSET_NULL;
  extern __shared__ uint32_t slds[];
  uint32_t* const sP = slds;
  uint32_t* const sT = slds + a.lds_words;
  const int lane = threadIdx.x;
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  const uint32_t w0 = FULL ? a.work_begin : 0u;
  const int max_records = FULL ? (int)min((long long)INT_MAX, a.pb_code_ints / (WI / 4)) : INT_MAX;
  const int thr2 = 2 * a.max_dist_thr;
  for (uint32_t wi = w0 + blockIdx.x; wi < w0 + nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    const WfaPairMeta pm = a.meta[pair];
    const int plen = pm.plen, tlen = pm.tlen;
    const int ak = tlen - plen;
    uint8_t* const rec = FULL ? reinterpret_cast<uint8_t*>(a.hist + (long long)(wi - w0) * a.hist_stride) : nullptr;   // this pair's history slot
    const uint32_t* gP = a.words + pm.p_woff;
    const uint32_t* gT = a.words + pm.t_woff;
    const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
    bool fallback = false;
    if (nwp + 3 > a.lds_words || nwt + 3 > a.lds_words || max_records <= 1) fallback = true;
    else {
      __syncthreads();
      for (int i = lane; i < nwp + 3; i += 64) sP[i] = (i < nwp) ? gP[i] : 0u;
      for (int i = lane; i < nwt + 3; i += 64) sT[i] = (i < nwt) ? gT[i] : 0u;
      __syncthreads();
    }
    int B = -(W / 2);  // diagonal of window position 0
    int result = 0, end_k = 0, end_off = 0, end_s = 0;
    int stop_status = 0, stop_score = 0;
    bool done = false;
    if (!fallback) {
      // per lane and chunk (doubled): kk2 = 2k, lim2 = 2 min(tlen, plen + k) (lim2c: not below 0), dlim2 = 2 max(tlen, plen + k)
      int kk2[NCH], lim2[NCH], lim2c[NCH], dlim2[NCH], cur[NCH], Mh[DM][NCH], Ih[E][NCH], Dh[E][NCH];
      uint32_t hoff[NCH];  // byte of this diagonal in the history record compute-next fills: (step + 1) * 128 + (k mod 128)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k = B + c * 64 + lane;
        kk2[c] = 2 * k; hoff[c] = WI + ((uint32_t)k & (WI - 1));
        lim2[c] = 2 * min(tlen, plen + k); lim2c[c] = max(lim2[c], 0); dlim2[c] = 2 * max(tlen, plen + k);
        cur[c] = (k == 0) ? 0 : NUL;  // wavefront 0
#pragma unroll
        for (int j = 0; j < E; ++j) { Ih[j][c] = NUL; Dh[j][c] = NUL; }
#pragma unroll
        for (int j = 0; j < DM; ++j) Mh[j][c] = NUL;
      }
      // neighbour registers of the 64-diagonal form: lane 0 (from below) / lane 63 (from above) never receive a value
      int nb_mo_lo = NUL, nb_ie_lo = NUL, nb_mo_hi = NUL, nb_de_hi = NUL;
      int step = 0, steps_wait = a.steps_between, dead_steps = 0;   // (the score of a step is step * g)
      const int min_wf_len_m1 = a.min_wf_len - 1;
      // the first step the loop must not start: the step limit reached (score step * g >= max_steps) or no room for the record it fills
      const int step_stop = (int)min(min((long long)(max_records - 1), ((long long)a.max_steps + a.g - 1) / a.g), 1ll << 24);
      int akp = ak - B;           // window position of the end diagonal
      int tlen2_eff = INT_MAX;    // 2 tlen while the end diagonal lies in an active chunk (the termination test reads lane akp & 63)
      bool big = true;
      int leave = 0;              // 1 reached the end, 2 step limit, 3 hand the pair on
#ifdef WFA_SLIM_COUNTERS
      uint32_t cnt_small = 0, cnt_big = 0, cnt_rounds = 0, cnt_cut = 0, cnt_shift = 0, cnt_oob = 0, cnt_live = 0;
#endif
      int togo = 0;               // steps left in the block of HP (an ending clears it)

      // One score step on ACT active chunks.  Every branch is a scalar branch and nothing leaves the step early: an ending sets
      // `leave` (1 reached the end, 2 step limit, 3 hand the pair on) and the rest of the step is skipped, so the loops around it
      // have one exit each.
      auto step_fn = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        // ---------------- extend M[s] (R/wavefront_extend_kernels.c:64-110) ----------------
        unsigned long long live[NCH] = {0ull, 0ull};
        unsigned long long keep[NCH] = {~0ull, ~0ull};   // lanes the cut-off keeps (all, unless it moves the wavefront's limits)
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
          bool more;
          do {
            more = false;
#ifdef WFA_SLIM_COUNTERS
            ++cnt_rounds;
#endif
#pragma unroll
            for (int c = 0; c < ACT; ++c) {
              const uint32_t m2 = min(slim_probe(sP, sT, v2[c], h2[c]), 64u);
              v2[c] += (int)m2; h2[c] += (int)m2;
              more |= (m2 == 64u) && (h2[c] < lim2[c]);   // (past the end the zero padding of both sequences would match on)
            }
          } while (__builtin_amdgcn_ballot_w64(more) != 0ull);
#pragma unroll
          for (int c = 0; c < ACT; ++c) {
            int cc = cur[c];
            asm("" : "+v"(cc));   // (a compare of its own: carried across the loop above, the first one's mask takes a round trip through a VGPR)
            const bool lv = cc >= 0;
            live[c] = __builtin_amdgcn_ballot_w64(lv);
            cur[c] = lv ? min(h2[c], lim2[c]) : cc;
          }
        }
        if (live[0] | live[1]) {
          dead_steps = 0;
#ifdef WFA_SLIM_COUNTERS
          cnt_live += (uint32_t)__builtin_popcountll(live[0]) + (uint32_t)__builtin_popcountll(live[1]);
#endif
          // ---------------- termination (R/wavefront_termination.c:37-61) ----------------
          int at_end;
          if (ACT == 1) at_end = __builtin_amdgcn_readlane(cur[0], akp & 63);
          else at_end = (akp & 64) ? __builtin_amdgcn_readlane(cur[1], akp & 63) : __builtin_amdgcn_readlane(cur[0], akp & 63);
          --steps_wait;
          if (at_end >= tlen2_eff) { leave = 1; togo = 0; }
          // ---------------- wf-adaptive cut-off (R/wavefront_heuristic.c:257-293, 509-567) ----------------
          else if (steps_wait <= 0) {
            const int lo_p = BD::first_pos(live), hi_p = BD::last_pos(live);   // window positions of the wavefront's ends
            if (hi_p - lo_p >= min_wf_len_m1) {
              int d[NCH], dm = 0x7fffffff;
#pragma unroll
              for (int c = 0; c < ACT; ++c) { d[c] = dlim2[c] - cur[c]; dm = min(dm, d[c]); }   // 2 max(plen - v, tlen - h); dead lanes ~ 2^30
              const int dmin = slim_wave_min(dm);   // (a live lane's distance never exceeds max(plen, tlen): the reference's initial value cannot win)
              unsigned long long ok[NCH] = {0ull, 0ull};
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
              // the wavefront's limits changed <=> a live lane lies outside them; then M, I, D are cut to them (the equate)
              if (ACT == 1) {
                const unsigned long long kp = (~0ull << new_lo) & (~0ull >> (63 - new_hi));
                if (live[0] & ~kp) keep[0] = kp;
#ifdef WFA_SLIM_COUNTERS
                if (live[0] & ~kp) ++cnt_cut;
#endif
              } else if (new_lo != lo_p || new_hi != hi_p) {
                // (positions 0..127 over two masks)
                const unsigned long long lo0 = (new_lo < 64) ? (~0ull << new_lo) : 0ull, lo1 = (new_lo < 64) ? ~0ull : (~0ull << (new_lo - 64));
                const unsigned long long hi0 = (new_hi < 64) ? (~0ull >> (63 - new_hi)) : ~0ull, hi1 = (new_hi < 64) ? 0ull : (~0ull >> (127 - new_hi));
                keep[0] = lo0 & hi0; keep[1] = lo1 & hi1;
              }
            }
          }
        } else {
          // nothing alive at this score: the first scores of the lattice; a ring that stays dead is left to the next stage
          if (++dead_steps > 2 * DM + 2) { leave = 3; togo = 0; }
        }
        if (leave == 0) {
          // (the cut, in each value's own register and without a branch: the paths of the step join on scalars only)
#pragma unroll
          for (int c = 0; c < ACT; ++c) { slim_keep(cur[c], keep[c]); slim_keep(Ih[0][c], keep[c]); slim_keep(Dh[0][c], keep[c]); }
          // ---------------- compute-next for score s + g (R/wavefront_compute_affine.c:44-86) ----------------
#pragma unroll
          for (int j = DM - 1; j > 0; --j)
#pragma unroll
            for (int c = 0; c < ACT; ++c) Mh[j][c] = Mh[j - 1][c];
#pragma unroll
          for (int c = 0; c < ACT; ++c) Mh[0][c] = cur[c];
          int ni[NCH], nd[NCH];
          unsigned long long oob = 0;
#pragma unroll
          for (int c = 0; c < ACT; ++c) {
            int mo_lo, ie_lo, mo_hi, de_hi;
            if (ACT == 1) {
              nb_mo_lo = __builtin_amdgcn_update_dpp(nb_mo_lo, Mh[OE - 1][0], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
              nb_ie_lo = __builtin_amdgcn_update_dpp(nb_ie_lo, Ih[E - 1][0], 0x138, 0xf, 0xf, false);
              nb_mo_hi = __builtin_amdgcn_update_dpp(nb_mo_hi, Mh[OE - 1][0], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
              nb_de_hi = __builtin_amdgcn_update_dpp(nb_de_hi, Dh[E - 1][0], 0x130, 0xf, 0xf, false);
              mo_lo = nb_mo_lo; ie_lo = nb_ie_lo; mo_hi = nb_mo_hi; de_hi = nb_de_hi;
            } else {
              mo_lo = BD::below(Mh[OE - 1], c); ie_lo = BD::below(Ih[E - 1], c);
              mo_hi = BD::above(Mh[OE - 1], c); de_hi = BD::above(Dh[E - 1], c);
            }
            ni[c] = max(mo_lo, ie_lo) + 2;
            nd[c] = max(mo_hi, de_hi);
            const int x1 = Mh[X - 1][c] + 2;
            const int t = max(ni[c], nd[c]);
            const int m = max(x1, t);
            if (FULL) {
              // the choice the backtrace would make (R/wavefront_backtrace.c:49-59: mismatch > deletion > insertion, extension >
              // opening on equal offsets), taken where the candidates are in registers
              // — the piggy-back history of score s + g: one byte per active diagonal, stored here (the window may move before the
              // next step begins; the record of score 0 is never read)
              // (BandArgs::pb_raw: the sign bits of four subtractions, shifted in one after the other)
              uint32_t cd = (uint32_t)(x1 - t) >> 31;
              cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(nd[c] - ni[c]), 31);
              cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(ie_lo - mo_lo), 31);
              cd = __builtin_amdgcn_alignbit(cd, (uint32_t)(de_hi - mo_hi), 31);
              rec[hoff[c]] = (uint8_t)cd;
              hoff[c] += WI;   // (an inactive chunk's offset is set again when it joins)
            }
            cur[c] = (m > lim2[c]) ? NUL : m;  // only M is clamped; negative values are dead already
            oob |= __builtin_amdgcn_ballot_w64(t > lim2[c]);
          }
#pragma unroll
          for (int j = E - 1; j > 0; --j)
#pragma unroll
            for (int c = 0; c < ACT; ++c) { Ih[j][c] = Ih[j - 1][c]; Dh[j][c] = Dh[j - 1][c]; }
#pragma unroll
          for (int c = 0; c < ACT; ++c) { Ih[0][c] = ni[c]; Dh[0][c] = nd[c]; }
#ifdef WFA_SLIM_COUNTERS
          if (oob) ++cnt_oob;
#endif
          if (oob) {
            // trim the ends of I and D (R/wavefront_compute.c:571-605): outside [first, last] in-bounds -> NULL
            unsigned long long bi[NCH] = {0ull, 0ull}, bd[NCH] = {0ull, 0ull};
#pragma unroll
            for (int c = 0; c < ACT; ++c) {
              bi[c] = __builtin_amdgcn_ballot_w64(ni[c] >= 0 && ni[c] <= lim2[c]);
              bd[c] = __builtin_amdgcn_ballot_w64(nd[c] >= 0 && nd[c] <= lim2[c]);
            }
            const int ilo = BD::first_pos(bi), ihi = BD::last_pos(bi), dlo = BD::first_pos(bd), dhi = BD::last_pos(bd);
#pragma unroll
            for (int c = 0; c < ACT; ++c) {
              const int p = c * 64 + lane;
              Ih[0][c] = (p < ilo || p > ihi) ? NUL : Ih[0][c];
              Dh[0][c] = (p < dlo || p > dhi) ? NUL : Dh[0][c];
            }
          }
          ++step;
          // ---------------- step limit (R/wavefront_unialign.c:98-107), room for the next step's history record ----------------
          if (step >= step_stop) { leave = ((long long)step * a.g >= (long long)a.max_steps) ? 2 : 3; togo = 0; }
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
#pragma unroll
            for (int j = 0; j < DM - 1; ++j) any &= Mh[j][c];   // (the oldest M is dropped by the next compute-next)
            hull[c] = __ballot(any >= 0);  // some register of this diagonal is not negative
          }
          if (!big) hull[1] = 0ull;   // (the inactive chunk's registers are stale, not read)
          const int fp = BD::first_pos(hull), lp = BD::last_pos(hull);
          if (lp >= 0) {
            const int width = lp - fp + 1;
            if (width > W - 20) leave = 3;
            // small form: the hull (and HP steps of growth either way) fits the first chunk; a little hysteresis keeps a hull near
            // the limit from being shifted to and fro
            const bool want_small = width <= (big ? WS - 2 * HP - 6 : WS - 2 * HP - 2);
            const int wa = want_small ? WS : W;
            if (fp < HP + 1 || lp > wa - HP - 2) {
#ifdef WFA_SLIM_COUNTERS
              ++cnt_shift;
#endif
              const int delta = fp - (wa - width) / 2;   // re-centre in the window (or in its small form)
              B += delta; akp -= delta;
              if (!big && want_small) {
                // 64 diagonals before and after: one lane shuffle per register
                const int src = lane + delta;
                const bool in = (unsigned)src < 64u;
                auto sh1 = [&](int& r) { const int t = __shfl(r, src & 63, 64); r = in ? t : NUL; };
                sh1(cur[0]);
#pragma unroll
                for (int j = 0; j < E; ++j) { sh1(Ih[j][0]); sh1(Dh[j][0]); }
#pragma unroll
                for (int j = 0; j < DM - 1; ++j) sh1(Mh[j][0]);
              } else {
                if (!big) {   // the stale registers of the inactive chunk read as NULL
                  cur[1] = NUL;
#pragma unroll
                  for (int j = 0; j < E; ++j) { Ih[j][1] = NUL; Dh[j][1] = NUL; }
#pragma unroll
                  for (int j = 0; j < DM; ++j) Mh[j][1] = NUL;
                }
                BD::shift(cur, delta, lane);
#pragma unroll
                for (int j = 0; j < E; ++j) { BD::shift(Ih[j], delta, lane); BD::shift(Dh[j], delta, lane); }
#pragma unroll
                for (int j = 0; j < DM - 1; ++j) BD::shift(Mh[j], delta, lane);
              }
#pragma unroll
              for (int c = 0; c < NCH; ++c) {
                const int k = B + c * 64 + lane;
                kk2[c] = 2 * k; hoff[c] = (uint32_t)(step + 1) * WI + ((uint32_t)k & (WI - 1));
                lim2[c] = 2 * min(tlen, plen + k); lim2c[c] = max(lim2[c], 0); dlim2[c] = 2 * max(tlen, plen + k);
              }
            } else if (!big && !want_small) {
              // small -> big without a shift: the inactive chunk joins as NULLs
              hoff[1] = (uint32_t)(step + 1) * WI + ((uint32_t)(B + 64 + lane) & (WI - 1));
              cur[1] = NUL;
#pragma unroll
              for (int j = 0; j < E; ++j) { Ih[j][1] = NUL; Dh[j][1] = NUL; }
#pragma unroll
              for (int j = 0; j < DM; ++j) Mh[j][1] = NUL;
            }
            big = !want_small;
          }
          tlen2_eff = (akp >= 0 && akp < (big ? W : WS)) ? 2 * tlen : INT_MAX;
        }
        togo = (leave == 0) ? HP : 0;
        if (!big) {
#pragma unroll 1
          while (togo > 0) { --togo; step_fn(band_int<1>{}); }
        } else {
#pragma unroll 1
          while (togo > 0) { --togo; step_fn(band_int<2>{}); }
        }
      }
#ifdef WFA_SLIM_COUNTERS
      if (a.dbg && lane == 0) {
        atomicAdd(a.dbg + 0, cnt_small); atomicAdd(a.dbg + 1, cnt_big); atomicAdd(a.dbg + 2, cnt_rounds); atomicAdd(a.dbg + 3, cnt_cut);
        atomicAdd(a.dbg + 4, cnt_shift); atomicAdd(a.dbg + 5, cnt_oob); atomicAdd(a.dbg + 6, cnt_live >> 6);
      }
#endif
      const int s_end = step * a.g;
      if (leave == 1) { done = true; result = -s_end; end_k = ak; end_off = tlen; end_s = s_end; }
      else if (leave == 2) { stop_status = WFA_STATUS_MAX_STEPS_REACHED; stop_score = -a.max_steps; }
      else fallback = true;
    }
    if (FULL && stop_status != 0 && lane == 0) {   // no end cell: no walk, empty op string (R/wavefront_unialign.c:147-237)
      a.cigar_begin[pair] = a.cigar_off[pair + 1];
      a.cigar_len[pair] = 0;
    }
    if (FULL && lane == 0) a.end_state[wi - w0] = make_int4(end_s, end_k, end_off, (fallback || stop_status != 0 || !done) ? 0 : 1);
    if (lane == 0) {
      if (fallback) {
        a.status[pair] = WFA_INTERNAL_FALLBACK;
        if (a.fb_list) a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;
      } else if (stop_status != 0) {
        a.score[pair] = stop_score;
        a.status[pair] = stop_status;
      } else {
        a.score[pair] = result;
        a.status[pair] = 0;
      }
    }
  }
}

template <bool FULL, int X, int OE, int E>
__global__ void __launch_bounds__(64)
wfa_slim_kernel(const BandArgs a) {
  wfa_slim_body<FULL, X, OE, E>(a);
}

#ifndef __HIPCC_RTC__
template <int X, int OE, int E>
static int launch_slim_shape(const BandArgs& a, bool full, long long grid, hipStream_t stream) {
  size_t smem = (size_t)a.lds_words * 2 * sizeof(uint32_t);
  static const int pad_kb = getenv("WFA_HIP_SLIM_LDS_PAD_KB") ? atoi(getenv("WFA_HIP_SLIM_LDS_PAD_KB")) : 0;   // (occupancy experiments)
  smem += (size_t)pad_kb << 10;
  if (full) hipLaunchKernelGGL((wfa_slim_kernel<true, X, OE, E>), dim3((unsigned)grid), dim3(64), smem, stream, a);
  else hipLaunchKernelGGL((wfa_slim_kernel<false, X, OE, E>), dim3((unsigned)grid), dim3(64), smem, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

}  // namespace wfa
