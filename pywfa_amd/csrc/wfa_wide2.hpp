// wfa_wide2.hpp — the wide-wavefront kernel, second form (round 3): gap-affine, rows of int16 offsets as
// in wfa_wide.hpp, but the pass works on PACKED pairs of diagonals and a step needs ONE barrier.
//
//   * A lane takes one 32-bit word of every row = two neighbouring diagonals (k, k + 1), and the recurrences of
//     R/wavefront_compute_affine.c:44-86 run on both halves at once with v_pk_max_i16 / v_pk_add_i16 (the lane kernel's
//     arithmetic, wfa_lane.hpp): four word loads (M[s-x], M[s-o-e], I[s-e], D[s-e]) instead of ten int16 loads, the k - 1 / k + 1
//     neighbours by one DPP wave shift + one v_alignbit each (the first / last lane of a wave reads its neighbour word itself).
//   * Only M is clamped (:80-84); the in-bounds tests that give the trimmed limits of M / I / D (R/wavefront_compute.c:571-605)
//     are sign bits of packed differences folded into two ballots per component.
//   * Extension (R/wavefront_extend_kernels.c:64-88): a 16-base first probe per half, then 32-base rounds for what runs on, one
//     run per lane at a time.
//   * One barrier per step: the scratch of a step (trim minima / maxima, end diagonal) is triple-buffered, every thread derives
//     the limits of the rows written in the previous step from that scratch itself, and the few gap cells outside their trimmed
//     limits are set to NULL behind a second barrier only in the steps that have any (wavefronts touching the matrix border).
//
// Scope: what wfa_wide_kernel<FULL, false, GROWS> covers (exact gap-affine, match = 0, 2-bit pairs, plen + tlen <= 32 000,
// end-to-end or ends-free, score-only or full CIGAR with the piggy-back history); same hand-over rules; gap-affine-2p and the
// int32 rows stay in wfa_wide.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include "wfa_common.hpp"
#include "wfa_hip.h"
#include "wfa_wide.hpp"

namespace wfa {

typedef short wide2_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t w2_max(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(wide2_s2, a), __builtin_bit_cast(wide2_s2, b)));
}
__device__ __forceinline__ uint32_t w2_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(wide2_s2, a) + __builtin_bit_cast(wide2_s2, b));
}
__device__ __forceinline__ uint32_t w2_sub(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(wide2_s2, a) - __builtin_bit_cast(wide2_s2, b));
}
#define WFA_WIDE2_NULL2 0xC000C000u   // (NULL, NULL) = (-16384, -16384)

template <bool FULL, bool GROWS>
__global__ void __launch_bounds__(1024)
wfa_wide2_kernel(const WideArgs a) {
  constexpr int NC = 3;
  extern __shared__ int wsm[];
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
  const int DM = max(a.X, a.OE);
  const int NM = DM + 1, NG1 = a.E + 1, NR = NM + 2 * NG1 + 1;
  const int rw = (a.wcap + 2 + 1) & ~1;            // halfs per row: guard, wcap diagonals, guard (+ pad); even: rows are word-aligned
  int* const ctrl = wsm;                           // scratch of step t at ctrl + 10 (t % 3): [0..2] trim min M / I / D, [3..5] max, [6] end k
  int* const rlo = wsm + WFA_WIDE_CTRL_INTS;       // trimmed limits of every row (written by thread 0 one step late, see above)
  int* const rhi = rlo + NR;
  uint32_t* const sP = reinterpret_cast<uint32_t*>(rhi + NR);
  uint32_t* const sT = sP + a.seq_words;
  short* rows;
  if constexpr (GROWS) rows = a.rows + (long long)blockIdx.x * a.rows_stride;
  else rows = reinterpret_cast<short*>(sT + a.seq_words);
  const int NULLROW = NR - 1;
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  int* const hist = FULL ? a.hist + (long long)blockIdx.x * a.hist_stride : nullptr;
  const uint32_t one2 = 0x00010001u;

  for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    const WfaPairMeta pm = a.meta[pair];
    const int plen = pm.plen, tlen = pm.tlen;
    const int ak = tlen - plen;
    const int pbf = a.ef ? a.pbf : 0, tbf = a.ef ? a.tbf : 0;
    // rows are centred between the start and the target diagonals; koff odd, so that diagonal k and k + 1 share a word iff k is odd...
    // (any parity works: word w holds the row indices 2 w and 2 w + 1)
    const int koff = a.wcap / 2 + 1 - (ak + tbf - pbf) / 2;     // row index of diagonal k = k + koff (1 .. wcap)
    const int kmin = 1 - koff, kmax = a.wcap - koff;
    __syncthreads();   // the previous pair is done with LDS and the rows
    {
      const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
      const uint32_t* gp = a.words + pm.p_woff;
      const uint32_t* gt = a.words + pm.t_woff;
      for (int i = tid; i < a.seq_words; i += T) { sP[i] = (i < nwp) ? gp[i] : 0u; sT[i] = (i < nwt) ? gt[i] : 0u; }
      uint32_t* r32 = reinterpret_cast<uint32_t*>(rows);
      const int n32 = NR * rw / 2;
      for (int i = tid; i < n32; i += T) r32[i] = WFA_WIDE2_NULL2;
      for (int i = tid; i < NR; i += T) { rlo[i] = 1; rhi[i] = -1; }
      if (tid < 30) ctrl[tid] = ((tid % 10) < 3 || (tid % 10) == 6) ? INT_MAX : INT_MIN;
    }
    bool hand_on = (plen + tlen > 32000) || (-pbf < kmin) || (tbf > kmax) || (ak < kmin) || (ak > kmax);
    int end_reason = 0;   // 1 reached, 3 handed on, 4 step limit
    int end_k = 0, end_t = 0;
    long long pb_used = 0;
    uint8_t* const pb_codes = FULL ? reinterpret_cast<uint8_t*>(hist) : nullptr;
    const long long pb_cap = FULL ? a.hist_stride * 4 : 0;
    int null_run = 0;
    // limits of the rows written in the previous step (kept in registers: thread 0 publishes them in rlo / rhi one step late)
    int pl_lo[NC] = {1, 1, 1}, pl_hi[NC] = {-1, -1, -1};
    __syncthreads();

    for (int t = 0; !hand_on; ++t) {
      const int s = t * a.g;
      int* const SC = ctrl + 10 * (t % 3);
      if (t > 0 && s >= a.max_steps) { end_reason = 4; break; }
      if (t > 16000) { end_reason = 3; break; }
      const int rM = t % NM, rI = NM + t % NG1, rD = NM + NG1 + t % NG1;
      const int iX = (t >= a.X) ? (t - a.X) % NM : NULLROW;
      const int iO = (t >= a.OE) ? (t - a.OE) % NM : NULLROW;
      const int iI = (t >= a.E) ? NM + (t - a.E) % NG1 : NULLROW;
      const int iD = (t >= a.E) ? NM + NG1 + (t - a.E) % NG1 : NULLROW;
      // limits of an input row: the rows of step t - 1 from the registers, older ones from rlo / rhi
      auto lim_lo = [&](int back, int comp, int row) { return (back == 1) ? pl_lo[comp] : rlo[row]; };
      auto lim_hi = [&](int back, int comp, int row) { return (back == 1) ? pl_hi[comp] : rhi[row]; };
      int lo, hi;
      if (t == 0) { lo = -pbf; hi = tbf; }
      else {
        const int xl = (t >= a.X) ? lim_lo(a.X, 0, iX) : 1, xh = (t >= a.X) ? lim_hi(a.X, 0, iX) : -1;
        const int ol = (t >= a.OE) ? lim_lo(a.OE, 0, iO) : 1, oh = (t >= a.OE) ? lim_hi(a.OE, 0, iO) : -1;
        const int il = (t >= a.E) ? lim_lo(a.E, 1, iI) : 1, ih = (t >= a.E) ? lim_hi(a.E, 1, iI) : -1;
        const int dl = (t >= a.E) ? lim_lo(a.E, 2, iD) : 1, dh = (t >= a.E) ? lim_hi(a.E, 2, iD) : -1;
        // R/wavefront_compute.c:40-86 (a null input counts with lo = 1, hi = -1, as there)
        lo = min(min(xl, ol - 1), min(il + 1, dl - 1));
        hi = max(max(xh, oh + 1), max(ih + 1, dh - 1));
        if (xl > xh && ol > oh && il > ih && dl > dh) { lo = 1; hi = -1; }
      }
      lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
      if (lo <= hi && (lo < kmin || hi > kmax)) { end_reason = 3; break; }
      if (lo > hi) { if (++null_run > DM + 2) { end_reason = 3; break; } } else null_run = 0;
      // codes of a step start on a word of the row (index (lo + koff) & ~1): the directory holds that diagonal
      const int lo_al = lo - ((lo + koff) & 1);
      long long code_base = 0;
      if (FULL) {
        const long long nb = (lo <= hi) ? (((long long)hi - lo_al + 3) & ~1ll) : 0;   // whole words of the row: two code bytes per word
        if (pb_used + nb + (long long)(t + 2) * 12 + 64 > pb_cap || pb_used + nb > 0x7fffff00ll) { end_reason = 3; break; }
        code_base = pb_used; pb_used += nb;
        if (tid == 0) { int* d = hist + a.hist_stride - 3ll * (t + 1); d[0] = (lo <= hi) ? lo_al : 1; d[1] = (lo <= hi) ? hi : 0; d[2] = (int)code_base; }
      }
      short* const wM = rows + (long long)rM * rw + koff;
      short* const wI = rows + (long long)rI * rw + koff;
      short* const wD = rows + (long long)rD * rw + koff;
      // stale cells of the rows written now (their previous wavefronts) outside the range written below
      {
        short* const wr[NC] = {wM, wI, wD};
        const int rws[NC] = {rM, rI, rD};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          // (the previous wavefront of this row: written NM resp. NG1 steps ago — never the previous step unless the ring has two rows)
          const int back = (c == 0) ? NM : NG1;
          const int olo = (back == 1) ? pl_lo[c] : rlo[rws[c]], ohi = (back == 1) ? pl_hi[c] : rhi[rws[c]];
          if (olo > ohi) continue;
          // untrimmed extent is not kept: NULL everything the old trimmed range covers outside the new range; cells the old pass
          // wrote beyond its trimmed limits were set to NULL then
          if (lo > hi) { for (int k = olo + tid; k <= ohi; k += T) wr[c][k] = (short)WFA_WIDE_NULL; continue; }
          for (int k = olo + tid; k <= min(ohi, lo - 1); k += T) wr[c][k] = (short)WFA_WIDE_NULL;
          for (int k = max(olo, hi + 1) + tid; k <= ohi; k += T) wr[c][k] = (short)WFA_WIDE_NULL;
        }
      }
      int wmin[NC], wmax[NC];   // (wave-uniform)
#pragma unroll
      for (int c = 0; c < NC; ++c) { wmin[c] = INT_MAX; wmax[c] = INT_MIN; }
      if (t == 0) {
        // wavefront 0 (R/wavefront_aligner.c:251-310): offset 0 on diagonal 0, the free begins on theirs; extended below like any M
        for (int k = lo + tid; k <= hi; k += T) wM[k] = (short)max(k, 0);
        __syncthreads();
      }
      if (lo <= hi) {
        const uint32_t* const x32 = reinterpret_cast<const uint32_t*>(rows + (long long)iX * rw);
        const uint32_t* const o32 = reinterpret_cast<const uint32_t*>(rows + (long long)iO * rw);
        const uint32_t* const i32 = reinterpret_cast<const uint32_t*>(rows + (long long)iI * rw);
        const uint32_t* const d32 = reinterpret_cast<const uint32_t*>(rows + (long long)iD * rw);
        uint32_t* const m32 = reinterpret_cast<uint32_t*>(rows + (long long)rM * rw);
        uint32_t* const ni32 = reinterpret_cast<uint32_t*>(rows + (long long)rI * rw);
        uint32_t* const nd32 = reinterpret_cast<uint32_t*>(rows + (long long)rD * rw);
        const int wlo = (lo + koff) >> 1, whi = (hi + koff) >> 1, wlast = rw / 2 - 1;
        for (int w0 = wlo + wave * 64; w0 <= whi; w0 += nwaves * 64) {
          const int w = w0 + lane;
          const bool act = w <= whi;
          const int wc = min(w, wlast);            // (idle lanes read their own word: it is the neighbour of the last active lane)
          const int kA = 2 * w - koff, kB = kA + 1;
          const bool inA = act && kA >= lo && kA <= hi, inB = act && kB >= lo && kB <= hi;
          uint32_t nm, ni, nd;
          uint32_t code2 = 0;
          if (t == 0) {
            nm = m32[wc]; ni = WFA_WIDE2_NULL2; nd = WFA_WIDE2_NULL2;
          } else {
            const uint32_t X = x32[wc], O = o32[wc], I = i32[wc], D = d32[wc];
            const uint32_t gi = w2_max(O, I), gd = w2_max(O, D);
            // the neighbour words: lane - 1 / lane + 1 of the wave, the wave's first / last lane from the rows
            uint32_t gi_prev = (uint32_t)__builtin_amdgcn_update_dpp((int)WFA_WIDE2_NULL2, (int)gi, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            uint32_t gd_next = (uint32_t)__builtin_amdgcn_update_dpp((int)WFA_WIDE2_NULL2, (int)gd, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            uint32_t ci_prev = 0, cd_next = 0, cmp_i = 0, cmp_d = 0;
            if (FULL) {
              cmp_i = w2_sub(I, O); cmp_d = w2_sub(D, O);   // sign: the extension is below the opening
              ci_prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cmp_i, 0x138, 0xf, 0xf, false);
              cd_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cmp_d, 0x130, 0xf, 0xf, false);
            }
            if (lane == 0) {
              const uint32_t Op = (wc > 0) ? o32[wc - 1] : WFA_WIDE2_NULL2, Ip = (wc > 0) ? i32[wc - 1] : WFA_WIDE2_NULL2;
              gi_prev = w2_max(Op, Ip);
              if (FULL) ci_prev = w2_sub(Ip, Op);
            }
            if (lane == 63) {
              const uint32_t On = (wc < wlast) ? o32[wc + 1] : WFA_WIDE2_NULL2, Dn = (wc < wlast) ? d32[wc + 1] : WFA_WIDE2_NULL2;
              gd_next = w2_max(On, Dn);
              if (FULL) cd_next = w2_sub(Dn, On);
            }
            // I(k) = max(M_oe, I_e)(k - 1) + 1, D(k) = max(M_oe, D_e)(k + 1): the max commutes with the shift
            ni = w2_add(__builtin_amdgcn_alignbit(gi, gi_prev, 16), one2);
            nd = __builtin_amdgcn_alignbit(gd_next, gd, 16);
            const uint32_t gap = w2_max(nd, ni), x1 = w2_add(X, one2);
            nm = w2_max(gap, x1);
            if (FULL) {
              // the comparison bits of wfa_lane.hpp: 8 mismatch below the best gap, 4 deletion below insertion, 2 / 1 the extension of
              // I / D below its opening; decoded into the origin codes of wfa_general.hpp PB below
              const uint32_t ca = w2_sub(x1, gap), cb = w2_sub(nd, ni);
              const uint32_t cc = __builtin_amdgcn_alignbit(cmp_i, ci_prev, 16);
              const uint32_t cd = __builtin_amdgcn_alignbit(cd_next, cmp_d, 16);
              // per half: mc = !A ? 0 : (!B ? 1 : 2); code = mc | (!C ? 4 : 0) | (!D ? 8 : 0)
              const uint32_t sa = (ca >> 15) & one2, sb = (cb >> 15) & one2, sc = (~cc >> 15) & one2, sd = (~cd >> 15) & one2;
              const uint32_t mc = (sa & ~sb) | ((sa & sb) << 1);
              const uint32_t cw = mc | (sc << 2) | (sd << 3);          // code of half A in bits 0-3, of half B in bits 16-19
              code2 = (cw & 0xFu) | ((cw >> 8) & 0xF00u);               // two bytes: diagonal kA, kB
            }
          }
          // cells outside [lo, hi] hold NULL
          const uint32_t rmask = (inA ? 0x0000ffffu : 0u) | (inB ? 0xffff0000u : 0u);
          ni = (ni & rmask) | (WFA_WIDE2_NULL2 & ~rmask);
          nd = (nd & rmask) | (WFA_WIDE2_NULL2 & ~rmask);
          nm = (nm & rmask) | (WFA_WIDE2_NULL2 & ~rmask);
          // in bounds: 0 <= offset <= min(tlen, plen + k) (offsets of live cells are never below max(k, 0)); only M is clamped
          const int limA = min(tlen, plen + kA), limB = min(tlen, plen + kB);
          const uint32_t lim2 = ((uint32_t)limA & 0xffffu) | ((uint32_t)limB << 16);   // (limA may be negative for k < -plen: nothing is in bounds there)
          const uint32_t om = nm | w2_sub(lim2, nm), oi = ni | w2_sub(lim2, ni), od = nd | w2_sub(lim2, nd);   // sign set: out of bounds
          {
            const uint32_t mk = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wide2_s2, om) >> (short)15);   // 0xffff per half out of bounds
            nm = (nm & ~mk) | (WFA_WIDE2_NULL2 & mk);
          }
          {
            const uint32_t oo[NC] = {om, oi, od};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const unsigned long long bA = __ballot(!(oo[c] & 0x8000u)), bB = __ballot(!(oo[c] & 0x80000000u));
              const unsigned long long any = bA | bB;
              if (any) {
                const int f = (int)__builtin_ctzll(any), l = 63 - (int)__builtin_clzll(any);
                wmin[c] = min(wmin[c], 2 * (w0 + f) - koff + (((bA >> f) & 1ull) ? 0 : 1));
                wmax[c] = max(wmax[c], 2 * (w0 + l) - koff + (((bB >> l) & 1ull) ? 1 : 0));
              }
            }
          }
          // ---- extend the two M cells (R/wavefront_extend_kernels.c:64-88) ----
          int hA = (int)(short)(nm & 0xffffu), hB = (int)nm >> 16;
          const bool liveA = hA >= 0, liveB = hB >= 0;
          if (__any(liveA || liveB)) {
            int leftA = liveA ? limA - hA : 0, leftB = liveB ? limB - hB : 0;
            bool moreA = false, moreB = false;
            {
              const int h = max(hA, 0), v = max(h - kA, 0);   // (dead cells probe position 0: their result is not used)
              const int pi = v >> 4, ti = h >> 4;
              const uint32_t x = __builtin_amdgcn_alignbit(sP[pi + 1], sP[pi], (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(sT[ti + 1], sT[ti], (uint32_t)h << 1);
              const int m = min((int)(wide_ffbl(x) >> 1), min(16, leftA));
              hA += liveA ? m : 0; leftA -= m;
              moreA = (m == 16) && (leftA > 0);
            }
            {
              const int h = max(hB, 0), v = max(h - kB, 0);
              const int pi = v >> 4, ti = h >> 4;
              const uint32_t x = __builtin_amdgcn_alignbit(sP[pi + 1], sP[pi], (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(sT[ti + 1], sT[ti], (uint32_t)h << 1);
              const int m = min((int)(wide_ffbl(x) >> 1), min(16, leftB));
              hB += liveB ? m : 0; leftB -= m;
              moreB = (m == 16) && (leftB > 0);
            }
            while (__any(moreA || moreB)) {
              // one run per lane at a time: A first, then B
              const bool selA = moreA;
              const bool run = moreA || moreB;
              int h = selA ? hA : hB, left = selA ? leftA : leftB;
              const int k = selA ? kA : kB;
              int v = h - k;
              if (!run) { h = 0; v = 0; left = 0; }   // (lanes without a run read position 0 and advance by 0)
              const int pi = v >> 4, ti = h >> 4;
              const uint32_t p0 = sP[pi], p1 = sP[pi + 1], p2 = sP[pi + 2], t0 = sT[ti], t1 = sT[ti + 1], t2 = sT[ti + 2];
              const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
              const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
              const uint32_t fb = min(wide_ffbl(xl), wide_ffbl(xh) | 32u);
              const int m = min((int)(fb >> 1), min(32, left));
              h += m; left -= m;
              const bool more = (m == 32) && (left > 0);
              if (run && selA) { hA = h; leftA = left; moreA = more; }
              else if (run) { hB = h; leftB = left; moreB = more; }
            }
            nm = ((uint32_t)hA & 0xffffu) | ((uint32_t)hB << 16);
            // termination on the extended offsets (R/wavefront_termination.c:37-61, :115-162: lowest k wins)
            if (a.ef) {
              if (liveA && ((hA >= tlen && plen - (hA - kA) <= a.pef) || (hA - kA >= plen && tlen - hA <= a.tef))) atomicMin(&SC[6], kA);
              if (liveB && ((hB >= tlen && plen - (hB - kB) <= a.pef) || (hB - kB >= plen && tlen - hB <= a.tef))) atomicMin(&SC[6], kB);
            } else {
              if (liveA && kA == ak && hA >= tlen) SC[6] = kA;
              if (liveB && kB == ak && hB >= tlen) SC[6] = kB;
            }
          }
          if (act) {
            // (negative gap values are stored as they are: they start at NULL and gain at most 1 per step, so they stay negative
            // for the 16 000 steps a pair may take here)
            m32[wc] = nm;
            if (t > 0) { ni32[wc] = ni; nd32[wc] = nd; }
            if (FULL && t > 0) *reinterpret_cast<uint16_t*>(pb_codes + code_base + (kA - lo_al)) = (uint16_t)code2;
          }
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) if (wmin[c] != INT_MAX) { atomicMin(&SC[c], wmin[c]); atomicMax(&SC[3 + c], wmax[c]); }
      }
      __syncthreads();   // rows, trimmed limits and the end flag of this step are visible
      // ---- trimmed limits (R/wavefront_compute.c:571-605): first / last in-bounds cell; none -> null ----
      int tlo[NC], thi[NC];
      bool need_null = false;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int mn = SC[c], mx = SC[3 + c];
        if (mn != INT_MAX) { tlo[c] = mn; thi[c] = mx; } else { tlo[c] = 1; thi[c] = -1; }
        if (c > 0 && lo <= hi && t > 0 && (tlo[c] > thi[c] || tlo[c] > lo || thi[c] < hi)) need_null = true;
      }
      const int ek = SC[6];
      if (ek != INT_MAX) { end_reason = 1; end_k = ek; end_t = t; }
      if (tid == 0) {
        const int rws[NC] = {rM, rI, rD};
        // (step 0 writes M only: its I / D rows stay null)
#pragma unroll
        for (int c = 0; c < NC; ++c) if (c == 0 || t > 0) { rlo[rws[c]] = tlo[c]; rhi[rws[c]] = thi[c]; }
        // the scratch of step t - 1 has been read for the last time before this barrier: reset it for step t + 2
        int* o = ctrl + 10 * ((t + 2) % 3);
        o[0] = INT_MAX; o[1] = INT_MAX; o[2] = INT_MAX; o[3] = INT_MIN; o[4] = INT_MIN; o[5] = INT_MIN; o[6] = INT_MAX;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) { pl_lo[c] = (c == 0 || t > 0) ? tlo[c] : 1; pl_hi[c] = (c == 0 || t > 0) ? thi[c] : -1; }
      if (end_reason) break;
      if (need_null) {
        // gap cells outside their trimmed limits become NULL (M's are NULL already); rare: wavefronts touching the matrix border
        short* const wr[NC] = {wM, wI, wD};
#pragma unroll
        for (int c = 1; c < NC; ++c) {
          const int l = tlo[c], h2 = thi[c];
          if (l > h2) { for (int k = lo + tid; k <= hi; k += T) wr[c][k] = (short)WFA_WIDE_NULL; }
          else {
            for (int k = lo + tid; k < l; k += T) wr[c][k] = (short)WFA_WIDE_NULL;
            for (int k = h2 + 1 + tid; k <= hi; k += T) wr[c][k] = (short)WFA_WIDE_NULL;
          }
        }
        __syncthreads();
      }
    }
    if (hand_on) end_reason = 3;
    __syncthreads();

    // =============================== finish (thread 0: as wfa_wide_kernel) ===============================
    if (tid == 0) {
      int out_score = 0, out_status = 0;
      long long cbeg = FULL ? a.cigar_off[pair + 1] : 0;
      int clen = 0;
      if (end_reason == 3) {
        out_status = WFA_INTERNAL_FALLBACK;
        a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;
      } else if (end_reason == 4) {
        out_status = WFA_STATUS_MAX_STEPS_REACHED; out_score = -a.max_steps;
      } else {
        out_score = -(end_t * a.g);
        if (FULL) {
          uint8_t* const ev = pb_codes + pb_used;
          const long long ev_cap = pb_cap - pb_used - (long long)(end_t + 2) * 12;
          int tc = end_t, k = end_k, comp = 0;
          long long nev = 0;
          bool fail = false;
          while (tc > 0) {
            if (nev >= ev_cap) { fail = true; break; }
            const int* d = hist + a.hist_stride - 3ll * (tc + 1);
            const int cd = (k >= d[0] && k <= d[1]) ? pb_codes[(long long)d[2] + (k - d[0])] : 0;
            const uint8_t flag = (comp == 0) ? 0x80 : 0;
            const int src = (comp == 0) ? ((cd & 3) == 0 ? 0 : ((cd & 3) == 1 ? 1 : 3)) : (comp == 1 ? 3 : 1);   // 0 mismatch, 1 D, 3 I
            if (src == 0) { ev[nev++] = (uint8_t)('X' | 0x80); tc -= a.X; }
            else if (src == 1) { ev[nev++] = (uint8_t)('D' | flag); ++k; if (cd & 8) { tc -= a.E; comp = 2; } else { tc -= a.OE; comp = 0; } }
            else { ev[nev++] = (uint8_t)('I' | flag); --k; if (cd & 4) { tc -= a.E; comp = 1; } else { tc -= a.OE; comp = 0; } }
          }
          if (fail || tc < 0) {
            out_status = WFA_INTERNAL_FALLBACK; out_score = 0;
            a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;
          } else {
            uint8_t* const out = a.cigar_ops + a.cigar_off[pair];
            long long n = 0;
            auto emit = [&](char c, int cnt) { for (int i = 0; i < cnt; ++i) out[n++] = (uint8_t)c; };
            auto lcp = [&](int v, int h) {
              const int maxrun = min(plen - v, tlen - h);
              int run = 0;
              while (run < maxrun) {
                const int pv = v + run, th = h + run;
                const uint32_t xp = __builtin_amdgcn_alignbit(sP[(pv >> 4) + 1], sP[pv >> 4], (uint32_t)(pv & 15) << 1);
                const uint32_t xt = __builtin_amdgcn_alignbit(sT[(th >> 4) + 1], sT[th >> 4], (uint32_t)(th & 15) << 1);
                const uint32_t x = xp ^ xt;
                const int m = x ? (__builtin_ctz(x) >> 1) : 16;
                run += m;
                if (m < 16) break;
              }
              return min(run, maxrun);
            };
            int h = max(k, 0), v = h - k;
            emit('I', h); emit('D', v);
            { const int e = lcp(v, h); emit('M', e); v += e; h += e; }
            for (long long e_ = nev - 1; e_ >= 0; --e_) {
              const int op = ev[e_] & 0x7F;
              if (op == 'X') { emit('X', 1); ++v; ++h; }
              else if (op == 'I') { emit('I', 1); ++h; }
              else { emit('D', 1); ++v; }
              if (ev[e_] & 0x80) { const int e = lcp(v, h); emit('M', e); v += e; h += e; }
            }
            emit('I', tlen - h); emit('D', plen - v);
            cbeg = a.cigar_off[pair];
            clen = (int)n;
          }
        }
      }
      a.score[pair] = out_score;
      a.status[pair] = out_status;
      if (FULL) { a.cigar_begin[pair] = cbeg; a.cigar_len[pair] = clen; }
    }
  }
}

}  // namespace wfa
