// wfa_seg.hpp — segmented register kernel: the C2 hot loop (gap-affine 4/6/2-shaped penalties, match 0,
// end-to-end, score only, no heuristic, reads <= 512 bases; same scope as the lane kernel, wfa_lane.hpp).
//
// Why segments.  On CDNA a VALU instruction occupies its SIMD for 4 cycles however few of the 64 lanes
// are live, and the one-alignment-per-wave kernel of round 1 already ran at the VALU issue limit
// (rocprofv3 SQ_INSTS_VALU x 4 cycles = kernel time).  At a few percent divergence a wavefront is a dozen
// diagonals wide, so lanes are the resource to share: the wave is cut into 64/W segments of W lanes
// (W = 8, 16, 32 or 64), each aligning its own pair inside a band of W diagonals, k in [c - W/2, c + W/2),
// centred between diagonal 0 and diagonal ak = tlen - plen: c = ceil(ak / 2).
//
// Exactness.  Cells outside the band are dropped, so the score found, S', is that of the best alignment
// staying inside the band (S' >= S, the reference's score).  It is accepted only when proven optimal: an
// end-to-end alignment that leaves the band has to climb from diagonal 0 to c + W/2 (or down to c - W/2 - 1)
// and come back to ak, which costs at least
//     Bmin = min(2o + e(2c + W - ak), 2o + e(W + 2 - 2c + ak))      (= 2o + eW or more for every |ak| < W),
// hence S' <= Bmin implies S' = S (score scope: a tie is the same score).  A pair that reaches Bmin without
// finishing is handed to the next stage (a wider segment, finally the banded / general kernels).
// In-bounds cells never descend from out-of-bounds ones (an I or D move keeps offset - lim), so for the
// score only M needs the clamp to lim; the reference's I/D end trimming (R/wavefront_compute.c:571-605)
// cannot change a score and is skipped.  Recurrences: R/wavefront_compute_affine.c:44-86; extension:
// R/wavefront_extend.c; termination: R/wavefront_termination.c (end2end).
//
// Work distribution.  Each wave owns one contiguous slice of the work list.  Pair metadata is held one
// window of 64 pairs per VGPR (lane i <-> pair i of the window, two windows so that prefetch can run
// ahead), a segment that finishes takes the next pair of the slice at once (its packed words were
// prefetched two pairs ahead into registers, then copied to the segment's LDS arrays), and results are
// stored straight from the lane that saw the end cell — segments never wait for each other.
//
// Extension: 32 bases per round and lane (three LDS words per sequence, two funnel shifts each, XOR, count
// trailing zeros), rounds repeated while any lane of the wave is still running.  (A variant in which a whole
// segment compares 16 W bases for its one long-running diagonal measured no faster and was dropped.)
#pragma once
#include "wfa_rtc_compat.hpp"
#include "wfa_common.hpp"
#include "wfa_hip.h"
#ifndef __HIPCC_RTC__
#include <string>
#include "wfa_rtc.hpp"
#endif

namespace wfa {

// index of the lowest set bit, ~0u for 0 (v_ffbl_b32 semantics)
__device__ __forceinline__ uint32_t ffbl_u32(uint32_t x) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// HEUR (round 3; score only, W <= 32, one-round extension): the general form for what the band bound cannot prove — wf-adaptive
// (R/wavefront_heuristic.c:257-293), free ends (R/wavefront_termination.c:115-162, wavefront 0 over the free begins).  The rule that
// keeps it exact is the lane kernel's (wfa_lane.hpp, HEUR): NO CLIPPING — a segment hands its pair on as soon as a cell of one of
// its two outermost lanes is alive, i.e. before the band has dropped anything the unbanded run would have kept.
// minimum over the W lanes of my segment, in every lane of it: butterflies inside the rows of 16 lanes by DPP (the move folds into the
// v_min), the two rows of a 32-lane segment by one ds_swizzle (lane ^ 16) — five ds_bpermute round trips before (round 4)
template <int W>
__device__ __forceinline__ int seg_min(int v) {
  static_assert(W == 8 || W == 16 || W == 32, "segments inside a 32-lane group");
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false));
  if (W >= 16) v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xf, 0xf, false));
  if (W == 32) v = min(v, __builtin_amdgcn_ds_swizzle(v, 0x401F /* bit mode: and 0x1f, or 0, xor 0x10 */));
  return v;
}
template <int W>
__device__ __forceinline__ int seg_max(int v) { return ~seg_min<W>(~v); }

// LIN: the one-component distances with CIGARs — no extension candidates (wfa_lane.hpp has the argument); run-time instantiation only
template <int X, int OE, int E, int W, bool LAZY, bool FULL, bool HEUR = false, int LIN = 0>
__global__ void __launch_bounds__(64)
wfa_seg_kernel(const FastArgs a) {
  static_assert(!(LAZY && FULL), "the history of a step is stored in the step itself");
  static_assert(!HEUR || (!LAZY && !FULL && W <= 32), "the general form: score only, every cell extended in its own step");
  static_assert(!LAZY || (X >= 2 && OE >= 2), "the lazy extension needs a wavefront to be consumed two steps after it is made (M is read at lags X and OE)");
  static_assert(W == 8 || W == 16 || W == 32 || W == 64, "segment width");
  constexpr int DM = (X > OE) ? X : OE;
  constexpr int NS = 64 / W, H = W / 2, LW = (W == 64) ? 6 : (W == 32) ? 5 : (W == 16) ? 4 : 3;
  constexpr unsigned long long FIELD = (W == 64) ? ~0ull : ((1ull << (W & 63)) - 1ull);
  constexpr int SW = WFA_FAST_WORDS;
  constexpr int NEVER = 0x7fffffff;
  __shared__ uint32_t lds[NS * 2 * SW + 4];  // per segment: pattern words, text words (32 + 2 each)
  __shared__ uint32_t ring[2][64];           // packed words of the next two pairs of the slice (direct-to-LDS loads)
  __shared__ uint32_t fbuf[64];              // pairs handed on, appended to the global list 64 at a time
  __shared__ uint32_t rpid[64];              // pairs finished: id and score, stored 64 at a time
  __shared__ int rscore[64];
  const int lane = threadIdx.x;
  const int seg = lane >> LW;
  const int l = lane & (W - 1);
  const int pbias = seg * 2 * SW * 16, tbias = pbias + SW * 16;  // base coordinates of my segment's words
  // my diagonal is k = l - H + c, c the centre of the pair's band (set when a pair is taken);
  // kb = pbias - k: pattern coordinate of offset x on my diagonal is x + kb
  int kb = pbias - (l - H);
  uint32_t nwork = __builtin_amdgcn_readfirstlane(a.nwork_dev ? *a.nwork_dev : a.nwork);  // keep everything derived from it scalar
  if (FULL && a.nwork_dev) {
    // leftovers of a previous stage, counted on the device: a.nwork history slots were reserved; what does not fit is
    // handed on unseen
    const uint32_t cap = a.nwork;
    if (nwork > cap) {
      for (uint32_t i = cap + blockIdx.x * 64u + threadIdx.x; i < nwork + 63u; i += gridDim.x * 64u) {
        const bool mine = i < nwork;
        const unsigned long long bm = __ballot(mine);
        uint32_t slot = 0;
        if (threadIdx.x == 0 && bm) slot = atomicAdd(a.fb_count, (uint32_t)__builtin_popcountll(bm));
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (mine) {
          const uint32_t pid = a.worklist[a.work_begin + i];
          a.fb_list[slot + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = pid;
          a.status[pid] = WFA_INTERNAL_FALLBACK;
        }
      }
      nwork = cap;
    }
  }
  const uint32_t per = __builtin_amdgcn_readfirstlane((nwork + gridDim.x - 1) / gridDim.x);  // (the division runs on the VALU)
  const unsigned long long begin64 = (unsigned long long)blockIdx.x * per;
  if (begin64 >= nwork) return;
  const uint32_t begin = (uint32_t)begin64;
  const uint32_t end = (uint32_t)min((unsigned long long)nwork, begin64 + per);
  // (words past the end of a staged sequence may hold anything: a run is clamped to the bases that remain)

  // No wave may wait for a global access it has just issued: taking a pair touches registers and LDS only.
  //  * metadata: two windows of 64 pairs in VGPRs (lane i holds pair wbase + i / wbase + 64 + i), read with
  //    v_readlane; the next window is requested at the end of a step, once per 64 pairs;
  //  * packed words: `global_load_lds_dword` straight into ring[], two pairs ahead (no VGPR, so no register
  //    copy that would have to wait for the load);
  //  * results: collected in LDS and stored 64 at a time.
  uint32_t pid0, pid1;
  WfaPairMeta m0, m1;
  auto load_window = [&](uint32_t wb, uint32_t& pid, WfaPairMeta& m) {
    const unsigned long long idx = (unsigned long long)wb + lane;
    pid = 0u; m.p_woff = 0u; m.t_woff = 0u; m.plen = 0; m.tlen = 0;
    if (idx < end) { pid = a.worklist ? a.worklist[a.work_begin + idx] : (uint32_t)(a.work_begin + idx); m = a.meta[pid]; }
  };
  uint32_t wbase = begin;
  load_window(wbase, pid0, m0);
  load_window(wbase + 64u, pid1, m1);
  // The text words of a pair follow its pattern words (csrc/wfa_hip.hip batch_build): one load of up to 64
  // consecutive words fetches both, lane j -> word j of the pair -> ring[slot][j].
  auto prefetch = [&](uint32_t i, uint32_t slot) {
    if (i < end) {
      const int r = (int)(i - wbase);
      uint32_t pw; int pl, tl;
      if (r < 64) {
        pw = __builtin_amdgcn_readlane(m0.p_woff, r);
        pl = __builtin_amdgcn_readlane(m0.plen, r); tl = __builtin_amdgcn_readlane(m0.tlen, r);
      } else {
        pw = __builtin_amdgcn_readlane(m1.p_woff, r - 64);
        pl = __builtin_amdgcn_readlane(m1.plen, r - 64); tl = __builtin_amdgcn_readlane(m1.tlen, r - 64);
      }
      const int ntot = ((pl + 15) >> 4) + ((tl + 15) >> 4);
      if (lane < ntot) __builtin_amdgcn_global_load_lds(a.words + pw + lane, &ring[slot][0], 4, 0, 0);
    }
  };
  uint32_t next_i = begin;
  prefetch(begin, 0u);
  prefetch(begin + 1u, 1u);
  uint32_t par = 0u;  // ring slot of pair next_i

  uint32_t nfb = 0;
  auto fb_flush = [&]() {
    if (nfb == 0u) return;
    __syncthreads();
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(a.fb_count, nfb);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if ((uint32_t)lane < nfb) { const uint32_t pid = fbuf[lane]; a.fb_list[slot + lane] = pid; a.status[pid] = WFA_INTERNAL_FALLBACK; }
    __syncthreads();
    nfb = 0u;
  };
  uint32_t nres = 0;
  auto res_flush = [&]() {
    if (nres == 0u) return;
    __syncthreads();
    if ((uint32_t)lane < nres) { const uint32_t pid = rpid[lane]; a.score[pid] = rscore[lane]; a.status[pid] = 0; }
    __syncthreads();
    nres = 0u;
  };

  // per-lane state of my segment's alignment.  The wave counts steps once (gstep); a pair taken at
  // gstep = s0 is at its own step gstep - s0 and must end by `deadline` = s0 + Bmin / g.
  int target = NEVER;  // tlen on the lane of diagonal tlen - plen: reaching it ends the alignment
  int lim = WFA_OFFSET_NULL, cur = WFA_OFFSET_NULL, s0 = 0, deadline = NEVER;
  uint32_t spair = 0;
  // FULL (round 4: piggy-back history, as the banded kernel's — one byte of origin codes per diagonal and step instead of an 8-byte
  // record of offsets: the history of a 64-lane segment was written at the rate of HBM): slot of my pair, my byte in the record of the
  // current step (position k mod W), and the code of my cell made by the last compute-next — origin of M (bits 0-1: 0 mismatch,
  // 1 deletion, 2 insertion), of I (bit 2: extension) and of D (bit 3); wfa_band_pb_bt_kernel walks it (a.seg_w = W)
  uint32_t tslot = 0;
  uint8_t* hp = nullptr;
  int code = 0;
  int Mh[DM], Ih[E], Dh[E];
#pragma unroll
  for (int d = 0; d < DM; ++d) Mh[d] = WFA_OFFSET_NULL;
#pragma unroll
  for (int d = 0; d < E; ++d) { Ih[d] = WFA_OFFSET_NULL; Dh[d] = WFA_OFFSET_NULL; }
  // HEUR: max(tlen, plen + k) of my diagonal (distance to the end = hdl - offset), lane of the end diagonal in my segment, the
  // distance of an empty wavefront, steps until the cut-off is looked at again, "a cell of an outermost lane is alive"
  int hdl = 0, hjt = 0, hdinit = 0, steps_wait = 0;
  int max_sw = 0; bool have_max_sw = false;   // X-drop: the largest cell score seen at a cut-off so far (segment-uniform)
  // X-drop (round 6): an alignment whose every cell was dropped ends "unreachable" once more than `scope` scores in a row had no input
  // (R/wavefront_unialign.c:262-265, R/wavefront_extend.c:97-104; wfa_band.hpp has the same count): last score with a non-null input,
  // and the score at which the count ran out (NEVER: it has not).  Before, such a pair sat in its segment until the cap on its steps
  // (4 (plen + tlen) + 64) and was then aligned again by the banded stage: 3 % of 150 bp pairs at 2 % under X-drop(100)
  int last_nonnull = 0, unreach_t = NEVER;
  bool edge = false;
  uint32_t want = (1u << NS) - 1u;  // segments waiting for a pair
  uint32_t busy = 0;                // segments aligning
  int gstep = 0;
  // LAZY: lanes whose newest cell (cur, wavefront s) / previous cell (Mh[0], wavefront s-1) still has bases to compare
  unsigned long long mcur = 0ull, mold = 0ull;

  // Terminates: gstep grows every round, every aligning segment has a finite deadline, and a segment only
  // takes a new pair while next_i < end.
  while (true) {
    if (want) {
      // (one wave per workgroup and the LDS serves a wave's instructions in order: no barrier is needed around the
      // staging writes, and __syncthreads() would also wait for the global loads just issued)
      do {
        const int s = __builtin_ctz(want);
        want &= want - 1u;
        if (next_i < end) {
          const uint32_t i = next_i++;
          const int r = (int)(i - wbase);
          int pl, tl; uint32_t pid;
          if (r < 64) {
            pl = __builtin_amdgcn_readlane(m0.plen, r); tl = __builtin_amdgcn_readlane(m0.tlen, r);
            pid = __builtin_amdgcn_readlane(pid0, r);
          } else {
            pl = __builtin_amdgcn_readlane(m1.plen, r - 64); tl = __builtin_amdgcn_readlane(m1.tlen, r - 64);
            pid = __builtin_amdgcn_readlane(pid1, r - 64);
          }
          // a pair this stage cannot take (too long, |tlen - plen| outside the band) is given an expired deadline:
          // the hand-over path below passes it on at once
          // The band is centred between diagonal 0 and diagonal ak = tlen - plen: c = ceil(ak / 2), k in [c - H, c + H).
          // HEUR: the band is centred on the span from the lowest to the highest diagonal the alignment must touch (the free begins and
          // the end diagonal), all of them at least one lane away from the segment's edges
          const int pbf_ = (HEUR && a.ef) ? a.pbf : 0, tbf_ = (HEUR && a.ef) ? a.tbf : 0;
          const int dlo = min(-pbf_, tl - pl), dhi = max(tbf_, tl - pl);
          const bool bad = pl > WFA_FAST_MAX_LEN || tl > WFA_FAST_MAX_LEN ||
                           (HEUR ? (dhi - dlo > W - 3 || pbf_ > pl || tbf_ > tl) : (tl - pl < 1 - 2 * H || tl - pl > 2 * H - 1));
          const int akk = bad ? 0x7fff : tl - pl;
          const int c = bad ? 0 : (HEUR ? ((dlo + dhi + 1) >> 1) : ((tl - pl + 1) >> 1));
          const int nwp = (pl + 15) >> 4, ntot = nwp + ((tl + 15) >> 4);
          // loads complete in order: only the load of pair i + 1 (the other slot), if there is one, may still be in flight
          if (i + 1u < end) __builtin_amdgcn_s_waitcnt(0xF71);  // vmcnt(1)
          else __builtin_amdgcn_s_waitcnt(0xF70);               // vmcnt(0)
          uint32_t w = ring[par][lane];
          asm volatile("" : "+v"(w));  // the slot has been read before it is refilled below
          if (!bad && lane < ntot) lds[s * 2 * SW + lane + ((lane >= nwp) ? SW - nwp : 0)] = w;
          prefetch(i + 2u, par);
          par ^= 1u;
          if (seg == s) {
            const int k = l - H + c;
            kb = pbias - k;
            target = (k == akk) ? tl : NEVER;
            lim = min(tl, pl + k);
            spair = pid; s0 = gstep;
            // Bmin / g in units of g (o / g = OE - E, e / g = E): leaving the band upwards costs a climb from 0 to
            // c + H and the way back to ak, downwards from 0 to c - H - 1 and back (LAZY: wavefront s is judged one
            // round later)
            deadline = bad ? gstep - 1
                           : HEUR ? gstep + 4 * (pl + tl) + 64   // (no bound to prove: only a cap on the steps a pair may take here)
                           : gstep + min(2 * (OE - E) + E * (2 * c + 2 * H - akk), 2 * (OE - E) + E * (2 * H + 2 - 2 * c + akk)) + (LAZY ? 1 : 0)
                                   - (FULL ? 1 : 0);  // FULL: S' < Bmin strictly, so that no co-optimal alignment leaves the band
            if (FULL) { tslot = i; hp = reinterpret_cast<uint8_t*>(a.hist + (long long)i * a.hist_stride) + (k & (W - 1)); code = 0; }
            cur = (bad || k != 0) ? WFA_OFFSET_NULL : 0;
            if (HEUR) {
              // wavefront 0 over the free begins (offset max(k, 0) on diagonals -pbf .. tbf); the threshold that ends the alignment on
              // my diagonal: end-to-end: offset tlen on the end diagonal; ends-free: h >= tlen with plen - v <= pef, or v >= plen with
              // tlen - h <= tef, i.e. offset >= min(max(tlen, plen + k - pef), max(plen + k, tlen - tef))
              if (!bad && a.ef && k >= -pbf_ && k <= tbf_) cur = max(k, 0);
              if (a.ef) target = bad ? NEVER : min(max(tl, pl + k - a.pef), max(pl + k, tl - a.tef));
              hdl = max(tl, pl + k); hjt = akk - (c - H); hdinit = max(pl, tl); steps_wait = a.steps_between; edge = false; have_max_sw = false;
              last_nonnull = 0; unreach_t = NEVER;
            }
#pragma unroll
            for (int d = 0; d < DM; ++d) Mh[d] = WFA_OFFSET_NULL;
#pragma unroll
            for (int d = 0; d < E; ++d) { Ih[d] = WFA_OFFSET_NULL; Dh[d] = WFA_OFFSET_NULL; }
          }
          if (LAZY) {
            const unsigned long long sm = FIELD << (s * W);
            mold &= ~sm;
            mcur = (mcur & ~sm) | (bad ? 0ull : (1ull << (s * W + H - c)));  // the cell (0, k = 0) is to be extended
          }
          busy |= 1u << s;
        }
      } while (want);
      if (!busy) break;
    }
    // ---------------- extend ----------------
    if (!LAZY) {
      // 32 bases per round on every diagonal until no lane is still running
      int left = (cur >= 0) ? lim - cur : 0;
      if (__any(left > 0)) {
        int h = max(cur, 0) + tbias, v = max(cur + kb, pbias);
        bool more;
        do {
          const int pi = v >> 4, ti = h >> 4;
          const uint32_t p0 = lds[pi], p1 = lds[pi + 1], p2 = lds[pi + 2];
          const uint32_t t0 = lds[ti], t1 = lds[ti + 1], t2 = lds[ti + 2];
          const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
          const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
          // first differing bit of xh:xl; v_ffbl_b32 returns ~0 for 0, so `| 32` is +32 or stays ~0, and the
          // unsigned min >> 1 is the number of equal bases or >= 2^30 when all 32 are equal
          const uint32_t fb = min(ffbl_u32(xl), ffbl_u32(xh) | 32u);
          const int m = min((int)(fb >> 1), min(32, left));
          v += m; h += m; left -= m;
          more = (m == 32) && (left > 0);
        } while (__any(more));
        if (cur >= 0) cur = h - tbias;
      }
    } else {
      // Wavefront s is first consumed when wavefront s + X is computed, so its extension may take two rounds of
      // the wave: one now and one in the next step, where the lane continues its previous cell (Mh[0]) instead of
      // starting its new one.  Only what is still running after that is waited for, so a long run of one
      // segment seldom holds up the other segments.
      if (mold | mcur) {
        do {
          const bool sel_old = __builtin_amdgcn_inverse_ballot_w64(mold);
          const bool sel_new = __builtin_amdgcn_inverse_ballot_w64(mcur & ~mold);
          int x = sel_old ? Mh[0] : (sel_new ? cur : W);  // (lanes without a job read in-range words and advance by 0)
          int left = (sel_old || sel_new) ? lim - x : 0;
          const int v = x + kb, h = x + tbias;
          const int pi = v >> 4, ti = h >> 4;
          const uint32_t p0 = lds[pi], p1 = lds[pi + 1], p2 = lds[pi + 2];
          const uint32_t t0 = lds[ti], t1 = lds[ti + 1], t2 = lds[ti + 2];
          const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
          const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
          const uint32_t fb = min(ffbl_u32(xl), ffbl_u32(xh) | 32u);
          const int m = min((int)(fb >> 1), min(32, left));  // left = 0: m = 0
          x += m; left -= m;
          const unsigned long long bm = __ballot(m == 32) & __ballot(left > 0);  // (two compares and an s_and: a combined bool goes through a VGPR)
          if (sel_old) Mh[0] = x;
          if (sel_new) cur = x;
          mcur &= mold | bm;  // a lane that served its previous cell has not started the new one yet
          mold &= bm;
        } while (mold);
      }
    }
    if (FULL) {
      // record of this step: the codes of its cells (step 0: none)
      if (deadline != NEVER) { *hp = (uint8_t)code; hp += W; }
    }
    // ---------------- termination / hand-over ----------------
    {
      const bool unr = HEUR && unreach_t != NEVER;          // segment-uniform: the null-step count ran out (no cell is alive)
      const bool rej = (gstep > deadline || (HEUR && edge)) && !unr;  // segment-uniform
      const bool fin = (LAZY ? Mh[0] : cur) >= target;  // possible on the lane of the end diagonal only (HEUR, free ends: on any lane)
      unsigned long long bfin = __ballot(fin);
      const unsigned long long brej = __ballot(rej);
      unsigned long long bunr = 0ull;
      if constexpr (HEUR) {
        bunr = __ballot(unr && l == 0);
        if (unr && l == 0) { a.score[spair] = -unreach_t; a.status[spair] = WFA_STATUS_PARTIAL; }   // (R/wavefront_unialign.c:147-237: no end cell, the score of the last step)
      }
      if (HEUR) {
        // one lane per segment reports (the score does not depend on which cell ended the alignment)
        unsigned long long one = 0ull;
#pragma unroll
        for (int s = 0; s < NS; ++s) { const unsigned long long f = bfin & (FIELD << (s * W)); one |= f & (0ull - f); }
        bfin = one;
      }
      const unsigned long long bd = bfin | brej | bunr;
      if (bd) {
        const unsigned long long ba = bfin & ~brej & ~bunr;
        const bool acc = __builtin_amdgcn_inverse_ballot_w64(ba);
        if (ba) {
          const uint32_t na = (uint32_t)__builtin_popcountll(ba);
          if (nres + na > 64u) res_flush();
          if (acc) {
            const uint32_t pos = nres + __builtin_amdgcn_mbcnt_hi((uint32_t)(ba >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ba, 0u));
            rpid[pos] = spair; rscore[pos] = -__mul24(gstep - s0 - (LAZY ? 1 : 0), a.g);
            if (FULL) a.end_state[tslot] = make_int4(__mul24(gstep - s0, a.g), pbias - kb, target, 1);  // {score, k = ak, offset = tlen}
          }
          nres += na;
        }
        const bool hand = rej && l == 0;
        const unsigned long long br = __ballot(hand);
        if (br) {
          const uint32_t nr = (uint32_t)__builtin_popcountll(br);
          if (nfb + nr > 64u) fb_flush();
          if (hand) {
            fbuf[nfb + __builtin_amdgcn_mbcnt_hi((uint32_t)(br >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)br, 0u))] = spair;
            if (FULL) a.end_state[tslot] = make_int4(0, 0, 0, 0);
          }
          nfb += nr;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
          if ((bd >> (s * W)) & FIELD) {
            want |= 1u << s; busy &= ~(1u << s);
            if (LAZY) { mcur &= ~(FIELD << (s * W)); mold &= ~(FIELD << (s * W)); }
          }
        if (((bd >> (seg * W)) & FIELD) != 0ull) { target = NEVER; deadline = NEVER; lim = WFA_OFFSET_NULL; cur = WFA_OFFSET_NULL; edge = false; unreach_t = NEVER; }
        if (!busy && next_i >= end) break;
      }
    }
    // ---------------- wf-adaptive cut-off (R/wavefront_heuristic.c:257-293, dispatcher :509-567) ----------------
    if constexpr (HEUR) {
      if (a.heur == 1) {
        const uint32_t f = (uint32_t)((__ballot(cur >= 0) >> (seg * W)) & FIELD);   // live lanes of my segment's M wavefront
        if (f != 0u) --steps_wait;   // (the cut-off is looked at only when the wavefront exists)
        const int lo = (int)__builtin_ctz(f | 0x80000000u), hi = 31 - (int)__builtin_clz(f | 1u);
        const bool consider = f != 0u && steps_wait <= 0 && (hi - lo + 1) >= a.min_wf_len && deadline != NEVER;
        if (__any(consider)) {
          // d = max(plen - v, tlen - h) = max(tlen, plen + k) - offset; dead lanes: far away
          const int d = (cur >= 0) ? hdl - cur : 0x3fffffff;
          int dmin = seg_min<(W > 32 ? 32 : W)>(d);
          dmin = min(dmin, hdinit);
          const uint32_t okf = (uint32_t)((__ballot(cur >= 0 && d - dmin <= a.max_dist_thr) >> (seg * W)) & FIELD);
          const int lc = okf ? (int)__builtin_ctz(okf) : 0x7fffffff, hc = okf ? 31 - (int)__builtin_clz(okf) : -0x7fffffff;
          int new_lo = lo, new_hi = hi;
          const int top_limit = min(hjt, hi);
          if (top_limit > lo) new_lo = min(lc, top_limit);
          const int bottom_limit = max(hjt, new_lo);
          if (bottom_limit < hi) new_hi = max(hc, bottom_limit);
          if (consider) {
            steps_wait = a.steps_between;
            // the dropped lanes read NULL in M, I and D from now on
            if ((new_lo != lo || new_hi != hi) && (l < new_lo || l > new_hi)) { cur = WFA_OFFSET_NULL; Ih[0] = WFA_OFFSET_NULL; Dh[0] = WFA_OFFSET_NULL; }
          }
        }
      } else if (a.heur == 2) {
        // X-drop (R/wavefront_heuristic.c:297-383; round 4): match = 0, so the "score" of a cell is (-(v + h) - s) / 2 (C division); the
        // cells more than xdrop below the largest score of the earlier cut-offs are dropped from both ends of this score's wavefront
        // (all of them, if none qualifies: the wavefronts of the other scores go on)
        const uint32_t f = (uint32_t)((__ballot(cur >= 0) >> (seg * W)) & FIELD);
        if (f != 0u) --steps_wait;
        const bool consider = f != 0u && steps_wait <= 0 && deadline != NEVER;
        if (__any(consider)) {
          const int lo = (int)__builtin_ctz(f | 0x80000000u), hi = 31 - (int)__builtin_clz(f | 1u);
          const int sw = (-(2 * max(cur, 0) - (pbias - kb)) - __mul24(gstep - s0, a.g)) / 2;   // (v + h = 2 offset - k)
          const int cmax = seg_max<(W > 32 ? 32 : W)>((cur >= 0) ? sw : -0x40000000);
          if (consider) {
            if (have_max_sw) {
              const uint32_t okf = (uint32_t)((__ballot(cur >= 0 && max_sw - sw < a.xdrop) >> (seg * W)) & FIELD);
              const int new_lo = okf ? (int)__builtin_ctz(okf) : hi + 1, new_hi = okf ? 31 - (int)__builtin_clz(okf) : hi;
              if ((new_lo != lo || new_hi != hi) && (l < new_lo || l > new_hi)) { cur = WFA_OFFSET_NULL; Ih[0] = WFA_OFFSET_NULL; Dh[0] = WFA_OFFSET_NULL; }
              if (cmax > max_sw) max_sw = cmax;
            } else {
              max_sw = cmax; have_max_sw = true;
            }
            steps_wait = a.steps_between;
          }
        }
      }
    }
    // ---------------- compute-next ----------------
#pragma unroll
    for (int d = DM - 1; d > 0; --d) Mh[d] = Mh[d - 1];
    Mh[0] = cur;
    {
      // I(k) = max(M_oe, I_e)(k-1) + 1 and D(k) = max(M_oe, D_e)(k+1): the max commutes with the lane shift
      const int mx = (LIN == 2) ? WFA_OFFSET_NULL : Mh[X - 1], mo = Mh[OE - 1], ie = LIN ? WFA_OFFSET_NULL : Ih[E - 1], de = LIN ? WFA_OFFSET_NULL : Dh[E - 1];
      if constexpr (HEUR) {
        if (a.heur == 2) {
          // the null-step count of the score being made: null = no input of my segment's wavefront holds an offset
          const bool nn = ((__ballot((mx & mo & ie & de) >= 0) >> (seg * W)) & FIELD) != 0ull;
          const int snew = __mul24(gstep + 1 - s0, a.g);
          if (deadline != NEVER && unreach_t == NEVER) {
            const int t_un = last_nonnull + a.scope + 1;   // (the scores between two multiples of g are null steps too)
            if (t_un < snew || (t_un == snew && !nn)) unreach_t = t_un;
            if (nn) last_nonnull = snew;
          }
        }
      }
      int ni, nd;
      if (FULL) {
        // the choices the backtrace would make (R/wavefront_backtrace.c:49-59: mismatch > deletion > insertion, extension > opening on
        // equal offsets), taken here where the candidates are in registers (as wfa_band.hpp, PB)
        const int mo_lo = seg_from_below<W>(mo), ie_lo = seg_from_below<W>(ie), mo_hi = seg_from_above<W>(mo), de_hi = seg_from_above<W>(de);
        ni = max(mo_lo, ie_lo) + 1;
        nd = max(mo_hi, de_hi);
        const int mc = (mx + 1 >= max(nd, ni)) ? 0 : ((nd >= ni) ? 1 : 2);
        code = mc | ((ie_lo >= mo_lo) ? 4 : 0) | ((de_hi >= mo_hi) ? 8 : 0);
      } else {
        ni = seg_from_below<W>(max(mo, ie)) + 1;
        nd = seg_from_above<W>(max(mo, de));
      }
      int nm = max(nd, max(mx + 1, ni));
      if (nm > lim) nm = WFA_OFFSET_NULL;
#pragma unroll
      for (int d = E - 1; d > 0; --d) { Ih[d] = Ih[d - 1]; Dh[d] = Dh[d - 1]; }
      Ih[0] = ni; Dh[0] = nd;
      cur = nm;
      if (LAZY) { mold = mcur; mcur = __ballot(nm >= 0); }
      if (HEUR) {
        // a cell of an outermost lane is alive (an offset or a gap value >= 0): the next steps could reach beyond the band
        const unsigned long long eb = __ballot((l == 0 || l == W - 1) && (nm & ni & nd) >= 0);
        edge = ((eb >> (seg * W)) & FIELD) != 0ull;
      }
    }
    ++gstep;
    // window 0 used up: window 1 moves down and the one after is requested (its use is >= 50 pairs away)
    if (next_i - wbase >= 64u) {
      pid0 = pid1; m0 = m1; wbase += 64u;
      load_window(wbase + 64u, pid1, m1);
    }
  }
  res_flush();
  fb_flush();
}

#ifndef __HIPCC_RTC__   // ---- host side (shape table, launch code) ----
// Penalty shapes (index, x, o + e, e) / gcd with an instantiation of the segmented kernel.  The first is pywfa's default
// 4/6/2 (also 2/3/1, 8/12/4, ...); the others are the usual short-read / long-read presets: 4/4/2, 4/6/1, 3/4/1,
// 6/5/3, 5/0/3 and unit costs 1/1/1.  Each shape is compiled in its own translation unit (csrc/k_seg.hip, once per
// index) so that the library builds in parallel; the entry points below are what the host side links against.
#define WFA_SEG_SHAPES(F) F(0, 2, 4, 1) F(1, 2, 3, 1) F(2, 4, 7, 1) F(3, 3, 5, 1) F(4, 6, 8, 3) F(5, 5, 3, 3) F(6, 1, 2, 1)

#define WFA_SEG_DECL(i, x, oe, e)                                                                        \
  int launch_seg_s##i(int w, bool lazy, unsigned grid, hipStream_t stream, const FastArgs& a);           \
  int launch_seg_full_s##i(int w, unsigned grid, hipStream_t stream, const FastArgs& a);                 \
  int launch_seg_heur_s##i(unsigned grid, hipStream_t stream, const FastArgs& a);
WFA_SEG_SHAPES(WFA_SEG_DECL)
#undef WFA_SEG_DECL

// index of the instantiated shape of these penalties; WFA_SHAPE_RTC for a shape the run-time path takes (c.rtc: hipRTC works here,
// csrc/wfa_rtc.cpp); -1 if none
inline int seg_shape(const WfaDevConfig& c, int* X, int* OE, int* E) {
  const int g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  *X = c.x / g; *OE = (c.o1 + c.e1) / g; *E = c.e1 / g;
#define WFA_SEG_MATCH(i, x, oe, e) if (*X == x && *OE == oe && *E == e && !(c.rtc && rtc_force_all())) return i;
  WFA_SEG_SHAPES(WFA_SEG_MATCH)
#undef WFA_SEG_MATCH
  if (c.rtc && rtc_shape_ok(*X, *OE, *E) && rtc_available()) return WFA_SHAPE_RTC;
  return -1;
}
// the segmented kernel of a run-time shape: wfa_seg_kernel<X, OE, E, W, LAZY, FULL, HEUR>
inline int launch_seg_rtc(int X, int OE, int E, int w, bool lazy, bool full, bool heur, unsigned grid, hipStream_t stream, const FastArgs& a) {
  if (lazy && (X < 2 || OE < 2)) lazy = false;   // (the two-round extension: M must not be read one step after it is made)
  const std::string name = "wfa::wfa_seg_kernel<" + std::to_string(X) + ", " + std::to_string(OE) + ", " + std::to_string(E) + ", " + std::to_string(w) + ", " +
                           rtc_bool(lazy) + ", " + rtc_bool(full) + ", " + rtc_bool(heur) + (a.lin ? ", " + std::to_string(a.lin) + ">" : std::string(">"));
  return rtc_launch("wfa_seg.hpp", name, grid, 64, 0, stream, &a, sizeof(a));
}

// which configurations the segmented kernels cover (score only, gap-affine, end-to-end or ends-free without free ends)
inline bool seg_supported(const WfaDevConfig& c, int ncomp, bool full) {
  if (full || ncomp != 3 || c.match != 0 || c.heuristic != 0 || c.wildcard >= 0) return false;
  if (c.endsfree && (c.pbf | c.pef | c.tbf | c.tef)) return false;
  if (c.max_steps != INT_MAX) return false;
  int X, OE, E;
  return seg_shape(c, &X, &OE, &E) >= 0;
}

template <int X, int OE, int E>
inline int launch_seg_shape(int w, bool lazy, unsigned grid, hipStream_t stream, const FastArgs& a) {
  const dim3 blk(64), g(grid);
  if constexpr (X >= 2 && OE >= 2) {
    if (lazy) {
      if (w == 8) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 8, true, false>), g, blk, 0, stream, a);
      else if (w == 32) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 32, true, false>), g, blk, 0, stream, a);
      else if (w == 64) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 64, true, false>), g, blk, 0, stream, a);
      else hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 16, true, false>), g, blk, 0, stream, a);
      return hipGetLastError() == hipSuccess ? 0 : -1;
    }
  }
  if (w == 8) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 8, false, false>), g, blk, 0, stream, a);
  else if (w == 32) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 32, false, false>), g, blk, 0, stream, a);
  else if (w == 64) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 64, false, false>), g, blk, 0, stream, a);
  else hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 16, false, false>), g, blk, 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the general score-only form (HEUR), segments of 32 lanes
template <int X, int OE, int E>
inline int launch_seg_heur_shape(unsigned grid, hipStream_t stream, const FastArgs& a) {
  hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 32, false, false, true>), dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// full-CIGAR launch with segments of `w` lanes: items [work_begin, work_begin + nwork) of the work list, one history slot
// each (a.nwork_dev set: the list is a previous stage's, a.nwork slots were reserved)
template <int X, int OE, int E>
inline int launch_seg_full_shape(int w, unsigned grid, hipStream_t stream, const FastArgs& a) {
  const dim3 g(grid);
  if (w == 32) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 32, false, true>), g, dim3(64), 0, stream, a);
  else if (w == 64) hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 64, false, true>), g, dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((wfa_seg_kernel<X, OE, E, 16, false, true>), g, dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// per_cu: slices of the work list per CU (one wave each).  8 times more slices than the 32 waves a CU holds: the SIMD
// issues oldest-first, so resident waves finish one after the other and a lone last wave cannot fill the VALU; with
// short slices the dispatcher refills the CU as waves retire (C2: 5.10 -> 4.42 ms).
inline int launch_seg_full(const WfaDevConfig& c, int cu_count, int per_cu, hipStream_t stream, FastArgs a, int w) {
  int X, OE, E;
  const int idx = seg_shape(c, &X, &OE, &E);
  if (idx < 0) return -1;
  a.g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  long long grid = std::min<long long>((long long)cu_count * per_cu, (long long)a.nwork);
  if (grid < 1) grid = 1;
  if (a.lin) return launch_seg_rtc(X, OE, E, (w == 32 || w == 64) ? w : 16, false, true, false, (unsigned)grid, stream, a);   // (run-time instantiation only)
#define WFA_SEG_LAUNCH_FULL(i, x, oe, e) if (idx == i) return launch_seg_full_s##i(w, (unsigned)grid, stream, a);
  WFA_SEG_SHAPES(WFA_SEG_LAUNCH_FULL)
#undef WFA_SEG_LAUNCH_FULL
  if (idx == WFA_SHAPE_RTC) return launch_seg_rtc(X, OE, E, (w == 32 || w == 64) ? w : 16, false, true, false, (unsigned)grid, stream, a);
  return -1;
}

// configurations of the general form of the 32-lane segments: what the lane kernel's general form takes (wfa_lane.hpp,
// lane_heur_config), without a step limit (the banded kernel reports the limit's status)
inline bool seg_heur_config(const WfaDevConfig& c, int ncomp) {
  int X, OE, E;
  return ncomp == 3 && c.match == 0 && c.wildcard < 0 && (c.heuristic == 0 || c.heuristic == 1 || c.heuristic == 2) && c.max_steps == INT_MAX &&
         seg_shape(c, &X, &OE, &E) >= 0;
}
// a.ef / a.pbf .. / a.heur .. set by the caller (FastArgs, wfa_common.hpp); the work list is a.worklist / a.nwork_dev / a.nwork
inline int launch_seg_heur(const WfaDevConfig& c, int cu_count, int per_cu, hipStream_t stream, FastArgs a) {
  int X, OE, E;
  const int idx = seg_shape(c, &X, &OE, &E);
  if (idx < 0) return -1;
  a.g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  a.hist = nullptr; a.hist_stride = 0; a.end_state = nullptr; a.work_begin = 0;
  long long grid = (long long)cu_count * per_cu;
  if (!a.nwork_dev && grid > (long long)a.nwork) grid = a.nwork;
  if (grid < 1) grid = 1;
#define WFA_SEG_LAUNCH_HEUR(i, x, oe, e) if (idx == i) return launch_seg_heur_s##i((unsigned)grid, stream, a);
  WFA_SEG_SHAPES(WFA_SEG_LAUNCH_HEUR)
#undef WFA_SEG_LAUNCH_HEUR
  if (idx == WFA_SHAPE_RTC) return launch_seg_rtc(X, OE, E, 32, false, false, true, (unsigned)grid, stream, a);
  return -1;
}

// steps a w-lane segment can take before it hands its pair on (+ 1), i.e. the records a history slot needs
// ints of a pair's history slot of the FULL form: the code records (w bytes each), the walk's event bytes, its run records
inline void seg_full_slot(const WfaDevConfig& c, int w, int max_len, long long* code_ints, long long* event_ints, long long* slot_ints);
inline int seg_full_records(const WfaDevConfig& c, int w) {
  int X, OE, E;
  if (seg_shape(c, &X, &OE, &E) < 0) return 0;
  return 2 * (OE - E) + E * (w + 1) + 3;
}

inline void seg_full_slot(const WfaDevConfig& c, int w, int max_len, long long* code_ints, long long* event_ints, long long* slot_ints) {
  *code_ints = (((long long)seg_full_records(c, w) * w + 15) & ~15ll) / 4;
  *event_ints = ((2ll * max_len + 16 + 15) & ~15ll) / 4;     // one event per edit
  *slot_ints = (*code_ints + *event_ints + 2ll * max_len + 8 + 15) & ~15ll;   // + one run record per op at most
}

// variant 6/7/8/9 = segments of 16/8/32/64 lanes (4/8/2/1 alignments per wave) with the two-round extension,
// 2/3/4/5 = the same widths extending every cell at once (the only form when x / g = 1)
inline int launch_seg(const WfaDevConfig& c, int cu_count, int per_cu, hipStream_t stream, const uint32_t* words,
                      const WfaPairMeta* meta, const uint32_t* worklist, const uint32_t* nwork_dev, uint32_t nwork,
                      int32_t* score, int32_t* status, uint32_t* fb_list, uint32_t* fb_count, int variant) {
  FastArgs a = FastArgs();   // (value-initialised: lin = 0 and every field no launch path sets)
  a.words = words; a.meta = meta; a.worklist = worklist; a.nwork_dev = nwork_dev; a.nwork = nwork;
  a.score = score; a.status = status; a.fb_list = fb_list; a.fb_count = fb_count;
  a.g = gcd_int(gcd_int(c.x, c.o1 + c.e1), c.e1);
  a.hist = nullptr; a.hist_stride = 0; a.end_state = nullptr; a.work_begin = 0;
  int X, OE, E;
  const int idx = seg_shape(c, &X, &OE, &E);
  if (idx < 0) return -1;
  long long grid = (long long)cu_count * per_cu;
  if (!nwork_dev && grid > (long long)nwork) grid = nwork;
  if (grid < 1) grid = 1;
  const bool lazy = variant >= 6;
  const int w = (variant == 3 || variant == 7) ? 8 : (variant == 4 || variant == 8) ? 32 : (variant == 5 || variant == 9) ? 64 : 16;
#define WFA_SEG_LAUNCH(i, x, oe, e) if (idx == i) return launch_seg_s##i(w, lazy, (unsigned)grid, stream, a);
  WFA_SEG_SHAPES(WFA_SEG_LAUNCH)
#undef WFA_SEG_LAUNCH
  if (idx == WFA_SHAPE_RTC) return launch_seg_rtc(X, OE, E, w, lazy, false, false, (unsigned)grid, stream, a);
  return -1;
}

#endif  // __HIPCC_RTC__

}  // namespace wfa
