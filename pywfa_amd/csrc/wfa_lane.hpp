// wfa_lane.hpp — lane-per-pair register kernel: the C2 hot loop of round 2 (gap-affine, match 0, end-to-end, score
// only, no heuristic, reads <= 512 bases; the scope of wfa_seg.hpp).
//
// Why.  Round 1's segmented kernel (wfa_seg.hpp: lane <-> diagonal, 4 pairs per wave) is latency-bound, not
// issue-bound: rocprofv3 on the 16-lane stage (profiles/r02_*) shows 7.4 resident waves per SIMD each issuing one
// instruction per ~14 cycles, the vector unit ~53 % busy, the scalar unit ~27 %: every step of a wave is one long
// dependent chain (address -> LDS -> funnel shift -> ffbl -> ballot -> branch) over the few live diagonals of 4 pairs.
// Here the layout is transposed: lane <-> PAIR (64 alignments per wave), and the W = 16 diagonals of a pair's band
// are 16 register slots unrolled in the instruction stream.  The 16 diagonals of a step are independent, so the wave
// carries 16 interleavable dependency chains and the unit of work per instruction is 64 pairs instead of 4:
//   * wavefront ring in VGPRs, two diagonals per register as int16 (offsets <= 512, NULL = -16384): M at depths
//     1..max(x, o+e), I and D at depth e, recurrences of R/wavefront_compute_affine.c:44-86 with v_pk_max_i16 /
//     v_pk_add_i16; the k-1 / k+1 neighbour is the adjacent half-register (one v_alignbit_b32 per register);
//   * extension (R/wavefront_extend_kernels.c:64-110): per slot 32 bases per probe from the lane's own 2-bit words in
//     LDS (three words per sequence, funnel shift, XOR, v_ffbl), repeated while any lane's run goes on; slots dead
//     in all 64 lanes are skipped with one ballot;
//   * termination (R/wavefront_termination.c:37-61) folded into the pass: one v_xad + v_min per slot;
//   * a lane that finishes stores its score and idles; once `refill_min` lanes idle they take the next pairs of
//     the wave's slice together: metadata by ds_bpermute from two 64-pair windows held in VGPRs, packed words by
//     one global_load_lds per pair straight into the lane's LDS slot, one s_waitcnt for all of them.
// Exactness is the band argument of wfa_seg.hpp, unchanged: band k in [c - 8, c + 8), c = ceil((tlen - plen) / 2);
// a score is kept only if S' <= Bmin = min(2o + e(2c + 16 - ak), 2o + e(18 - 2c + ak)), otherwise the pair is handed
// to the next stage (32- / 64-lane segments, banded, general kernel) — every stage computes the same wavefronts.
//
// HEUR (round 3; score only): the same layout for what the band bound cannot prove — wf-adaptive (R/wavefront_heuristic.c:257-293),
// ends-free spans with free ends (R/wavefront_aligner.c:259-302, R/wavefront_termination.c:115-162) and a step limit
// (R/wavefront_unialign.c:98-107).  A heuristic result is not the optimum, so "S' <= Bmin" proves nothing; instead the band must
// hold the WHOLE wavefront: the pair is handed on the moment a cell of its outermost slots (0 or 15) comes alive — until then no
// cell outside the band can exist, and everything computed equals the unbanded computation, cut-offs included.  The cut-off per
// pair: live slots by sign bits, distances max(plen - v, tlen - h) on packed halves, first / last slot within max_distance of the
// best, the dropped slots set to NULL in M, I and D (the equate, :161-172).
//
// FULL (round 3): full CIGARs of short reads in the same layout, with the piggy-back history of the long-read kernels
// (R/wavefront_backtrace_offload.c:39-73, R/wavefront_pcigar.c:204-266): compute-next also records, per cell, which
// candidate the backtrace would take (R/wavefront_backtrace.c:49-59: mismatch > deletion > insertion on equal offsets, extension
// > opening) as four comparison bits — 8 bytes per lane and step for the 16 diagonals, and a wave-step's 64 x 8 bytes leave
// as ONE coalesced 512-byte store into the wave's record list in HBM.  A pair that ends leaves {record of its end step, lane, end
// slot, steps}; wfa_lane_walk_kernel (wfa_band.hpp; one thread per alignment, every lane busy) follows the bits back — one 8-byte
// load per edit —, unpacks forwards re-extending the matches on the packed words (a wavefront cell is always extended to its end)
// and leaves a handful of run records {length, op}; wfa_lane_expand_kernel turns the runs into op bytes.  96 MB of codes per
// million pairs instead of round 2's 1.7 GB of 8-byte offset records.  The bound is applied strictly (S' < Bmin) as in
// wfa_seg_kernel<.., FULL>, so that every candidate the reference's backtrace compares lies inside the band with its true value.
// (First form of this round: records in LDS and the walk inside this kernel at every refill — 12 of 64 lanes busy in the walk,
// 2.25 waves per SIMD: 0.80 ms per million pairs against 0.30 for the score alone; dropped.)
#pragma once
#include "wfa_rtc_compat.hpp"
#include "wfa_common.hpp"
#include "wfa_hip.h"
#ifndef __HIPCC_RTC__
#include "wfa_rtc.hpp"
#endif

namespace wfa {

typedef short lane_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short lane_u2 __attribute__((ext_vector_type(2)));

#define WFA_LANE_NULL16 (-16384)
#define WFA_LANE_NULL2 0xC000C000u
#ifndef WFA_LANE_DEBUG_COUNTERS
#define WFA_LANE_DEBUG_COUNTERS 0  // 1: a.hist (if set) receives eight uint64: {wave-steps, refills, parked runs, parked 32-base rounds,
#endif                             //    first-probe blocks, rounds of second runs, hand-over blocks, second runs} (score-only form)
// Analysis builds (tools/lane_mix.py): region marks as comments in the assembly, so that the instructions of each region of the
// step can be counted by class and weighted with the counters above (the dynamic instruction mix of the kernel)
#ifndef WFA_LANE_REGION_MARKS
#define WFA_LANE_REGION_MARKS 0
#endif
#if WFA_LANE_REGION_MARKS
#define WFA_LANE_MARK(name) asm volatile("; WFA_MARK " name ::: "memory")
#else
#define WFA_LANE_MARK(name)
#endif
#define WFA_LANE_COUNT(i) do { if (WFA_LANE_DEBUG_COUNTERS && !FULL && a.hist) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.hist) + (i), 1ull); } } while (0)

__device__ __forceinline__ uint32_t lane_ffbl(uint32_t x) {  // index of the lowest set bit, ~0u for 0
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  lane_s2 r = __builtin_elementwise_max(__builtin_bit_cast(lane_s2, a), __builtin_bit_cast(lane_s2, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
  lane_s2 r = __builtin_elementwise_min(__builtin_bit_cast(lane_s2, a), __builtin_bit_cast(lane_s2, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
  lane_s2 r = __builtin_bit_cast(lane_s2, a) - __builtin_bit_cast(lane_s2, b);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
  lane_s2 r = __builtin_bit_cast(lane_s2, a) + __builtin_bit_cast(lane_s2, b);
  return __builtin_bit_cast(uint32_t, r);
}
// per half: nm > lim ? NULL : nm   (only M is clamped, R/wavefront_compute_affine.c:80-84; negative values are dead anyway)
__device__ __forceinline__ uint32_t pk_clamp(uint32_t nm, uint32_t lim) {
  lane_s2 d = __builtin_bit_cast(lane_s2, lim) - __builtin_bit_cast(lane_s2, nm);
  d = d >> (short)15;  // 0xffff where nm > lim
  const uint32_t m = __builtin_bit_cast(uint32_t, d);
  return (nm & ~m) | (WFA_LANE_NULL2 & m);
}

#ifndef WFA_LANE_WAVES_PER_EU
#define WFA_LANE_WAVES_PER_EU 0   // > 0: cap the registers for that many waves per SIMD (the 4/6/2 shape needs 109 VGPRs: 4 waves)
#endif
#if WFA_LANE_WAVES_PER_EU > 0
#define WFA_LANE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(WFA_LANE_WAVES_PER_EU, WFA_LANE_WAVES_PER_EU)))
#else
#define WFA_LANE_OCCUPANCY
#endif

// records of origin codes a lane can need: steps 0 .. Bmin / g - 1 (wfa_seg.hpp: Bmin / g <= 2 (OE - E) + E (2 H + 1))
template <int OE, int E>
struct LaneFull { static constexpr int NREC = 2 * (OE - E) + E * 17 + 1; };

// LIN (round 6): the one-component distances with CIGARs (gap-linear, levenshtein: R/wavefront_compute_linear.c:44-74, R/wavefront_compute_edit.c:
// 44-100, R/wavefront_backtrace.c:223-319) as gap-affine with o = 0 WITHOUT the extension candidates: I(k) = M_e(k - 1) + 1, D(k) = M_e(k + 1)
// are then exactly the one-component recurrences, and the origin codes say "opened" for every gap — the walk never enters a gap component,
// which is the linear backtrace (the same priority: mismatch > deletion > insertion).  With the extension candidates the values are the
// same but ties between an opening and an extension go to the extension (R/wavefront_backtrace.c:49-59), and the walk is then bound to
// the gap component where the linear backtrace would be free to take a mismatch: different op strings.  Instantiated at run time only.
// LIN = 2: indel — the same without the mismatch candidate (R/wavefront_compute_edit.c:44-100 with the indel metric).
// NRP (round 6, HEUR only): packed registers per component — 8 (16 diagonals) or 16: a band of 32 diagonals for pairs whose whole
// wavefront the 16 slots cannot hold (150 bp at 2 %: 45 % of the pairs pass score 20, where the hull reaches slots 0 / 15; 6 % pass 36).
template <int X, int OE, int E, bool FULL, bool HEUR = false, int LIN = 0, int NRP = 8>
__global__ void __launch_bounds__(64) WFA_LANE_OCCUPANCY
wfa_lane_kernel(const FastArgs a, const int slot_words_seq, const int refill_arg) {
  static_assert(!(FULL && HEUR), "the general form is score only");
  static_assert(NRP == 8 || (HEUR && NRP == 16), "the wider band exists in the general form only");
  const int refill_min = refill_arg & 0xff;
  constexpr int NR = NRP, W = 2 * NR, H = NR;     // band of 16 diagonals = 8 packed registers (NRP = 16: 32 diagonals)
  constexpr int DM = (X > OE) ? X : OE;           // depth of the M ring
  constexpr int NEVER = 0x7fffffff;
  constexpr int NREC = LaneFull<OE, E>::NREC;     // FULL: steps a pair can take here (bounds the walk)
  extern __shared__ uint32_t lds[];               // [4 guard words][64 slots x slot_words][4 guard words]
  const int slot_words = slot_words_seq;
  const int lane = threadIdx.x;
  uint32_t nwork = __builtin_amdgcn_readfirstlane(a.nwork_dev ? *a.nwork_dev : a.nwork);
  // the wave's slice of the work list: a fixed one (blockIdx.x-th of gridDim.x), or — score-only forms with a.dyn_next — chunks taken
  // from a device counter until the list is used up
  const bool dyn = !FULL && a.dyn_next != nullptr;
  bool exhausted = !dyn;
  uint32_t begin, end;
  auto grab = [&](uint32_t& b_, uint32_t& e_) -> bool {
    uint32_t nb = 0;
    if (lane == 0) nb = atomicAdd(a.dyn_next, a.dyn_chunk);
    nb = __builtin_amdgcn_readfirstlane(nb);
    if (nb >= nwork) return false;
    b_ = nb; e_ = (uint32_t)min((unsigned long long)nwork, (unsigned long long)nb + a.dyn_chunk);
    return true;
  };
  if (dyn) {
    if (!grab(begin, end)) return;
  } else {
    const uint32_t per = __builtin_amdgcn_readfirstlane((nwork + gridDim.x - 1) / gridDim.x);
    const unsigned long long begin64 = (unsigned long long)blockIdx.x * per;
    if (begin64 >= nwork) return;
    begin = (uint32_t)begin64;
    end = (uint32_t)min((unsigned long long)nwork, begin64 + per);
  }

  // ---- two windows of 64 pairs' metadata: lane i holds pair wbase + i / wbase + 64 + i
  uint32_t pid0, pw0, ln0, pid1, pw1, ln1;        // ln = plen | tlen << 16 (0xffffffff: too long for this stage)
  auto load_window = [&](uint32_t wb, uint32_t& pid, uint32_t& pw, uint32_t& ln) {
    const unsigned long long idx = (unsigned long long)wb + lane;
    pid = 0u; pw = 0u; ln = 0u;
    if (idx < end) {
      pid = a.worklist ? a.worklist[a.work_begin + idx] : (uint32_t)(a.work_begin + idx);
      const WfaPairMeta m = a.meta[pid];
      pw = m.p_woff;
      ln = (m.plen > WFA_FAST_MAX_LEN || m.tlen > WFA_FAST_MAX_LEN) ? 0xffffffffu : ((uint32_t)m.plen | ((uint32_t)m.tlen << 16));
    }
  };
  uint32_t wbase = begin, next_i = begin;
  load_window(wbase, pid0, pw0, ln0);
  load_window(wbase + 64u, pid1, pw1, ln1);

  // ---- per-lane state
  const int pbase = (4 + lane * slot_words) * 16;  // base coordinate (in bases) of my LDS slot
  uint32_t cur[NR], lim[NR], Mh[DM][NR], Ih[E][NR], Dh[E][NR];
  uint32_t thr[HEUR ? NR : 1];   // HEUR: termination thresholds per slot, two per register
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    cur[r] = WFA_LANE_NULL2; lim[r] = WFA_LANE_NULL2;
    if (HEUR) thr[r] = 0x3fff3fffu;
#pragma unroll
    for (int d = 0; d < DM; ++d) Mh[d][r] = WFA_LANE_NULL2;
#pragma unroll
    for (int d = 0; d < E; ++d) { Ih[d][r] = WFA_LANE_NULL2; Dh[d][r] = WFA_LANE_NULL2; }
  }
  int kb0 = pbase, tb = pbase;   // pattern coordinate of offset x on slot j is x + kb0 - j; text coordinate is x + tb
  int jt = 0;                    // slot of the end diagonal tlen - plen
  uint32_t tend = 0xffffu;       // tlen: the offset that ends the alignment on slot jt (0xffff: no pair)
  int s0 = 0, deadline = NEVER;
  // HEUR: termination thresholds per slot (an offset >= thr ends the alignment; 0x3fff: never), the packed plen + k and tlen
  // of my pair, the cut-off countdown
  uint32_t my_lb2 = 0, my_tl2 = 0;
  int steps_wait = 0, my_dinit = 0;
  bool edge_live = false;
  uint32_t mypid = 0;
  unsigned long long idle = ~0ull;  // lanes without a pair
  int gstep = 0;
  // FULL: my history slot (work item index of this launch); the wave's record list; once it is full the wave records nothing
  // more and hands on what finishes afterwards
  uint32_t myslot = 0;
  uint2* const wave_codes = FULL ? a.codes + (unsigned long long)blockIdx.x * (unsigned long long)a.codes_cap * 64ull : nullptr;
  bool codes_full = false;
  // (round 6) what the wave hands on is collected in LDS and leaves 64 or more at a time: ONE atomic on the list's counter per flush.  An
  // atomic per hand-over event serialises on that one address — the 16-diagonal general form hands on 11 % of 150 bp pairs at 1 %
  // divergence and took 1.55 ms per 2 M pairs where the 32-diagonal form took 1.06 ms
  __shared__ uint32_t rejq[128];
  int nrej = 0;
  auto rej_flush = [&]() {
    if (nrej == 0) return;
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(a.fb_count, (uint32_t)nrej);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (lane < nrej) a.fb_list[slot + lane] = rejq[lane];
    if (lane + 64 < nrej) a.fb_list[slot + 64 + lane] = rejq[64 + lane];
    nrej = 0;
  };

  while (true) {
    WFA_LANE_MARK("looptop_begin");   // (analysis builds: what precedes the first of these is the prologue)
    // =================== take pairs ===================
    const int nidle = __builtin_popcountll(idle);
    if (!exhausted && next_i >= end && (nidle >= refill_min || idle == ~0ull)) {
      // this slice is used up: the next chunk of the list, if any (its metadata windows replace the old ones: every pair of the old
      // slice has been taken)
      if (grab(begin, end)) { wbase = begin; next_i = begin; load_window(wbase, pid0, pw0, ln0); load_window(wbase + 64u, pid1, pw1, ln1); }
      else exhausted = true;
    }
    if (next_i < end && (nidle >= refill_min || idle == ~0ull)) {
      WFA_LANE_MARK("refill_begin");
      WFA_LANE_COUNT(1);
      const bool is_idle = __builtin_amdgcn_inverse_ballot_w64(idle);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      const uint32_t navail = end - next_i;
      const bool take = is_idle && rank < navail;
      const uint32_t src = next_i + rank - wbase;   // 0 .. 126: position in the two windows
      const int sl = (int)(src & 63u);
      const uint32_t a_pid = __shfl(pid0, sl, 64), a_pw = __shfl(pw0, sl, 64), a_ln = __shfl(ln0, sl, 64);
      const uint32_t b_pid = __shfl(pid1, sl, 64), b_pw = __shfl(pw1, sl, 64), b_ln = __shfl(ln1, sl, 64);
      const bool hi_w = src >= 64u;
      const uint32_t n_pid = hi_w ? b_pid : a_pid, n_pw = hi_w ? b_pw : a_pw, n_ln = hi_w ? b_ln : a_ln;
      const int pl = (int)(n_ln & 0xffffu), tl = (int)(n_ln >> 16);
      const int nwp = (pl + 15) >> 4, ntot = nwp + ((tl + 15) >> 4);
      const int ak = tl - pl;
      // a pair this stage cannot take (too long for the slot, |tlen - plen| outside the band) gets an expired deadline:
      // the hand-over path below passes it on at once
      // HEUR: the band is centred on the span from the lowest to the highest diagonal the alignment must touch (the free begins
      // and the end diagonal), all of them at least one slot away from the edges
      const int pbf_ = (HEUR && a.ef) ? a.pbf : 0, tbf_ = (HEUR && a.ef) ? a.tbf : 0;
      const int dlo = min(-pbf_, ak), dhi = max(tbf_, ak);
      const bool bad = n_ln == 0xffffffffu || ntot + 1 > slot_words_seq ||
                       (HEUR ? (dhi - dlo > 2 * H - 3) : (ak < 1 - 2 * H || ak > 2 * H - 1));
      const unsigned long long tmask = __ballot(take);
      // packed words of every taken pair: one direct-to-LDS load each (lane j -> word j of the pair -> slot word j;
      // the text words of a pair follow its pattern words, csrc/wfa_hip.hip batch_build)
      {
        unsigned long long lm = __ballot(take && !bad);
        while (lm) {
          const int L = __builtin_ctzll(lm);
          lm &= lm - 1ull;
          const uint32_t pw = __builtin_amdgcn_readlane(n_pw, L);
          const int nt = __builtin_amdgcn_readlane(ntot, L);
          if (lane < nt) __builtin_amdgcn_global_load_lds(a.words + pw + lane, &lds[4 + L * slot_words], 4, 0, 0);
        }
      }
      if (take) {
        const int c = bad ? 0 : (HEUR ? ((dlo + dhi + 1) >> 1) : ((ak + 1) >> 1));      // band centre: k in [c - H, c + H)
        const int k0 = c - H;                         // diagonal of slot 0
        mypid = n_pid; s0 = gstep;
        if (FULL) myslot = next_i + rank;   // (slot = index of the work item in this launch)
        kb0 = pbase - k0; tb = pbase + nwp * 16;
        jt = bad ? 0 : ak - k0;
        tend = bad ? 0xffffu : (uint32_t)tl;
        // Bmin / g in units of g (o / g = OE - E, e / g = E), see wfa_seg.hpp
        deadline = bad ? gstep - 1
                       : HEUR ? gstep + 4 * (pl + tl) + 64   // (no bound to prove: only a cap on the steps a pair may take here)
                       : gstep + min(2 * (OE - E) + E * (2 * c + 2 * H - ak), 2 * (OE - E) + E * (2 * H + 2 - 2 * c + ak))
                               - (FULL ? 1 : 0);   // FULL: S' < Bmin strictly, so that no co-optimal alignment leaves the band
        const int j0 = -k0;                           // slot of diagonal 0: the cell (score 0, offset 0)
        // lim of slot j = min(tlen, plen + k0 + j), two per register (a pair this stage cannot take: NULL, nothing lives)
        const uint32_t lb2 = ((uint32_t)(pl + k0) & 0xffffu) | ((uint32_t)(pl + k0 + 1) << 16);
        const uint32_t tl2 = (uint32_t)tl | ((uint32_t)tl << 16);
        const int j0r = bad ? -1 : (j0 >> 1);
        const uint32_t c0 = (j0 & 1) ? ((uint32_t)WFA_LANE_NULL16 & 0xffffu) : (WFA_LANE_NULL2 & 0xffff0000u);  // offset 0 in half j0 & 1
        if (HEUR) { my_lb2 = lb2; my_tl2 = tl2; steps_wait = a.steps_between; my_dinit = max(pl, tl); }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          lim[r] = bad ? WFA_LANE_NULL2 : pk_min(tl2, pk_add(lb2, (uint32_t)(2 * r) * 0x00010001u));
          cur[r] = (j0r == r) ? c0 : WFA_LANE_NULL2;
          if (HEUR) {
            // wavefront 0 over the free begins (offset max(k, 0) on diagonals -pbf .. tbf) and the thresholds that end the alignment:
            // end-to-end: offset tlen on the end diagonal; ends-free: h >= tlen with plen - v <= pef, or v >= plen with tlen - h <= tef,
            // i.e. offset >= min(max(tlen, plen + k - pef), max(plen + k, tlen - tef))
            uint32_t c2 = 0, t2 = 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int k = k0 + 2 * r + q;
              int w0v = (k == 0) ? 0 : WFA_LANE_NULL16;
              if (a.ef && k >= -pbf_ && k <= tbf_) w0v = max(k, 0);
              int th = (k == ak) ? tl : 0x3fff;
              if (a.ef) th = min(max(tl, pl + k - a.pef), max(pl + k, tl - a.tef));
              if (bad) { w0v = WFA_LANE_NULL16; th = 0x3fff; }
              th = max(0, min(th, 0x3fff));
              c2 |= ((uint32_t)w0v & 0xffffu) << (16 * q);
              t2 |= ((uint32_t)th & 0xffffu) << (16 * q);
            }
            cur[r] = c2; thr[r] = t2;
          }
#pragma unroll
          for (int d = 0; d < DM; ++d) Mh[d][r] = WFA_LANE_NULL2;
#pragma unroll
          for (int d = 0; d < E; ++d) { Ih[d][r] = WFA_LANE_NULL2; Dh[d][r] = WFA_LANE_NULL2; }
        }
      }
      const uint32_t ntake = min((uint32_t)nidle, navail);
      next_i += ntake;
      idle &= ~tmask;
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the words are in LDS
      asm volatile("" ::: "memory");
      if (next_i - wbase >= 64u) {           // window 0 used up: window 1 moves down, the one after is requested
        pid0 = pid1; pw0 = pw1; ln0 = ln1; wbase += 64u;
        load_window(wbase + 64u, pid1, pw1, ln1);
      }
      WFA_LANE_MARK("refill_end");
    } else if (idle == ~0ull && exhausted) {
      rej_flush();
      break;                                 // nothing left
    }

    // =================== extend (all 16 slots) ===================
    // One packed register (two slots) at a time: the first probes (16 bases: two words per sequence) of its two
    // slots are straight-line code; a register whose slots are dead in all 64 lanes is skipped after a test on the
    // packed values (at a few percent divergence two or three of the eight registers are live).  A run that goes on (the pair's true diagonal, mostly) becomes the lane's ONE pending
    // run {slot, offset, bases left}; all lanes' pending runs are then continued together, 32 bases per round, so the
    // longest run of the 64 pairs is waited for once per step and not once per slot.  (A second run of the same
    // lane in the same step — rare — is finished on the spot.)
    int hot_j = -1, hot_x = 0, hot_left = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      // some slot of this register holds a cell (offset >= 0) that can still run (offset < lim): per half min(cur, lim - 1 - cur) >= 0
      const uint32_t ua = pk_min(cur[r], pk_add(pk_sub(lim[r], cur[r]), 0xffffffffu));
      if (__any((ua & 0x80008000u) != 0x80008000u)) {
        WFA_LANE_MARK("probe_begin");
        WFA_LANE_COUNT(4);
        int off[2], left[2], x[2];
        bool valid[2], more[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          off[q] = q ? ((int)cur[r] >> 16) : (int)(short)(cur[r] & 0xffffu);
          const int lj = q ? ((int)lim[r] >> 16) : (int)(short)(lim[r] & 0xffffu);
          valid[q] = off[q] >= 0;
          left[q] = valid[q] ? lj - off[q] : 0;   // longest possible run; in-bounds cells have 0 <= off <= lim
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int j = 2 * r + q;
          x[q] = max(off[q], 0);
          const int v = x[q] + kb0 - j, h = x[q] + tb;
          const uint32_t pa = ((uint32_t)v >> 2) & ~3u, ta = ((uint32_t)h >> 2) & ~3u;  // byte address of the word holding base v / h
          const uint32_t* pp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + pa);
          const uint32_t* tp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + ta);
          const uint32_t p0 = pp[0], p1 = pp[1], t0 = tp[0], t1 = tp[1];
          const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
          const int m = min((int)(lane_ffbl(xl) >> 1), min(16, left[q]));   // (ffbl of 0 is ~0: all 16 equal)
          x[q] += m; left[q] -= m;
          more[q] = (m == 16) && (left[q] > 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int j = 2 * r + q;
          // the first run that goes on is parked; a lane that already has one finishes this one now
          const bool park = more[q] && hot_j < 0;
          const bool now = more[q] && !park;
          hot_x = park ? x[q] : hot_x; hot_left = park ? left[q] : hot_left; hot_j = park ? j : hot_j;
          if (__any(now)) {
            WFA_LANE_MARK("nowfix_begin");
            WFA_LANE_COUNT(7);
            bool mo = now;
            int lf = mo ? left[q] : 0;       // other lanes advance by 0
            int xx = x[q];
            do {
              WFA_LANE_MARK("now_begin");
              WFA_LANE_COUNT(5);
              const int v = xx + kb0 - j, h = xx + tb;
              const uint32_t pa = ((uint32_t)v >> 2) & ~3u, ta = ((uint32_t)h >> 2) & ~3u;
              const uint32_t* pp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + pa);
              const uint32_t* tp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + ta);
              const uint32_t p0 = pp[0], p1 = pp[1], p2 = pp[2], t0 = tp[0], t1 = tp[1], t2 = tp[2];
              const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
              const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
              const uint32_t fb = min(lane_ffbl(xl), lane_ffbl(xh) | 32u);   // first differing bit of xh:xl (~0 if none)
              const int m = min((int)(fb >> 1), min(32, lf));
              xx += m; lf -= m;
              mo = (m == 32) && (lf > 0);
              WFA_LANE_MARK("now_end");
            } while (__any(mo));
            x[q] = xx;
            WFA_LANE_MARK("nowfix_end");
          }
        }
        x[0] = valid[0] ? x[0] : off[0]; x[1] = valid[1] ? x[1] : off[1];
        cur[r] = ((uint32_t)x[0] & 0xffffu) | ((uint32_t)x[1] << 16);
        WFA_LANE_MARK("probe_end");
      }
    }
    if (__any(hot_j >= 0)) {
      // the parked runs, all lanes together
      bool mo = hot_j >= 0;
      int lf = mo ? hot_left : 0;
      const int kbj = kb0 - hot_j;
      WFA_LANE_COUNT(2);
      WFA_LANE_MARK("parkfix_begin");
      do {
        WFA_LANE_MARK("parked_begin");
        const int v = hot_x + kbj, h = hot_x + tb;
        const uint32_t pa = ((uint32_t)v >> 2) & ~3u, ta = ((uint32_t)h >> 2) & ~3u;
        const uint32_t* pp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + pa);
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + ta);
        const uint32_t p0 = pp[0], p1 = pp[1], p2 = pp[2], t0 = tp[0], t1 = tp[1], t2 = tp[2];
        const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
        const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
        const uint32_t fb = min(lane_ffbl(xl), lane_ffbl(xh) | 32u);
        const int m = min((int)(fb >> 1), min(32, lf));
        hot_x += m; lf -= m;
        mo = (m == 32) && (lf > 0);
        WFA_LANE_COUNT(3);
        WFA_LANE_MARK("parked_end");
      } while (__any(mo));
      // back into its half register (lanes without a parked run: hot_j = -1 matches no register)
      const int hr = hot_j >> 1;
      const uint32_t hmask = (hot_j & 1) ? 0xffff0000u : 0x0000ffffu;
      const uint32_t hval = (hot_j & 1) ? ((uint32_t)hot_x << 16) : ((uint32_t)hot_x & 0xffffu);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const uint32_t mk = (hr == r) ? hmask : 0u;
        cur[r] = (cur[r] & ~mk) | (hval & mk);
      }
      WFA_LANE_MARK("parkfix_end");
    }

    // =================== termination / hand-over ===================
    {
      // R/wavefront_termination.c:37-61: the cell of the end diagonal (slot jt) has reached offset tlen
      uint32_t endv = 0;
      bool hit = false;
      if constexpr (HEUR) {
        // some slot has reached its threshold (ends-free: R/wavefront_termination.c:115-162; the score does not depend on which)
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) acc |= ~pk_sub(cur[r], thr[r]);   // sign clear: offset >= threshold
        hit = (acc & 0x80008000u) != 0u;
      } else {
        uint32_t sel = cur[0];
#pragma unroll
        for (int r = 1; r < NR; ++r) sel = ((jt >> 1) == r) ? cur[r] : sel;
        endv = (jt & 1) ? (sel >> 16) : (sel & 0xffffu);
      }
      const unsigned long long active = ~idle;
      const unsigned long long bfin = __ballot(HEUR ? hit : (endv == tend)) & active;
      // HEUR: also the step limit (tested after compute-next of a score, before its extension: the first score >= max_steps lies
      // at or before the next step's, R/wavefront_unialign.c:98-107) and a pair whose band no longer holds its wavefront
      unsigned long long blimit = 0ull;
      if constexpr (HEUR) blimit = __ballot(a.max_steps != NEVER && __mul24(gstep - s0 + 1, a.g) >= a.max_steps) & active & ~bfin;
      const unsigned long long brej = __ballot(gstep > deadline || (HEUR && edge_live) || (FULL && codes_full)) & active & ~blimit;
      const unsigned long long bd = bfin | brej | blimit;
      if (bd) {
        WFA_LANE_MARK("bd_begin");
        WFA_LANE_COUNT(6);
        const unsigned long long ba = bfin & ~brej;
        if (__builtin_amdgcn_inverse_ballot_w64(ba)) {
          a.score[mypid] = -__mul24(gstep - s0, a.g);
          a.status[mypid] = 0;
          // FULL: where the walk starts: the record of this step in the wave's list, my lane and end slot, the steps taken
          if (FULL) a.end_state[myslot] = make_int4((int)(blockIdx.x * (unsigned)a.codes_cap + (unsigned)gstep), lane | (jt << 8), gstep - s0, 1);
        }
        if (HEUR && __builtin_amdgcn_inverse_ballot_w64(blimit)) {
          a.score[mypid] = -a.max_steps;
          a.status[mypid] = WFA_STATUS_MAX_STEPS_REACHED;
        }
        if (brej) {
          if (__builtin_amdgcn_inverse_ballot_w64(brej)) {
            rejq[nrej + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(brej >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)brej, 0u))] = mypid;
            a.status[mypid] = WFA_INTERNAL_FALLBACK;
            if (FULL) a.end_state[myslot] = make_int4(0, 0, 0, 0);
          }
          nrej += __builtin_popcountll(brej);
          if (nrej >= 64) rej_flush();
        }
        if (__builtin_amdgcn_inverse_ballot_w64(bd)) {
          // an idle lane computes nothing that lives: every new cell is clamped away
#pragma unroll
          for (int r = 0; r < NR; ++r) { cur[r] = WFA_LANE_NULL2; lim[r] = WFA_LANE_NULL2; if (HEUR) thr[r] = 0x3fff3fffu; }
          deadline = NEVER; jt = 0; tend = 0xffffu; edge_live = false;
        }
        idle |= bd;
        WFA_LANE_MARK("bd_end");
        if (idle == ~0ull && next_i >= end && exhausted) { rej_flush(); break; }   // (a flush behind the loop costs 40 registers)
      }
    }

    // =================== wf-adaptive cut-off (R/wavefront_heuristic.c:257-293, dispatcher :509-567) ===================
    if constexpr (HEUR) {
      if (a.heur == 1) {
        // live slots of M as a bit mask per pair (bit j: slot j holds an offset).  Two instructions per register: the sign bits as 0 / 1
        // halves (v_pk_lshrrev_b16), then a dot product with the two bit weights of the register accumulates them (v_dot2_u32_u16; eight
        // registers = 16 bits per accumulator)
        auto mask16 = [&](const uint32_t (&x)[NR]) -> uint32_t {   // bit 2 r + q = sign bit of half q of x[r] is CLEAR
          uint32_t lo16 = 0, hi16 = 0;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const lane_u2 t = __builtin_bit_cast(lane_u2, x[r]) >> (unsigned short)15;
            const uint32_t w = (1u << (2 * (r & 7))) | (1u << (2 * (r & 7) + 17));
            if (r < 8) lo16 = __builtin_amdgcn_udot2(t, __builtin_bit_cast(lane_u2, w), lo16, false);
            else hi16 = __builtin_amdgcn_udot2(t, __builtin_bit_cast(lane_u2, w), hi16, false);
          }
          return ~(lo16 | (hi16 << 16)) & (NR == 16 ? 0xffffffffu : 0xffffu);
        };
        const uint32_t live = mask16(cur);
        if (live != 0u) --steps_wait;   // (the cut-off is looked at only when the wavefront exists)
        const int lo = (int)lane_ffbl(live), hi = 31 - (int)__builtin_clz(live | 1u);   // (live == 0: lo = -1 as unsigned: width test fails)
        const bool consider = live != 0u && steps_wait <= 0 && (hi - lo + 1) >= a.min_wf_len && !__builtin_amdgcn_inverse_ballot_w64(idle);
        if (__any(consider)) {
          // d = max(plen - v, tlen - h) = max(tlen, plen + k) - offset; dead slots: far away
          uint32_t d2[NR];
          uint32_t dm = 0x3fff3fffu;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const uint32_t dl = pk_max(my_tl2, pk_add(my_lb2, (uint32_t)(2 * r) * 0x00010001u));
            const uint32_t dv = pk_sub(dl, cur[r]);
            const uint32_t dead = __builtin_bit_cast(uint32_t, __builtin_bit_cast(lane_s2, cur[r]) >> (short)15);   // 0xffff per dead half
            d2[r] = (dv & ~dead) | (0x3fff3fffu & dead);
            dm = pk_min(dm, d2[r]);
          }
          const int dmin = min(my_dinit, min((int)(dm & 0xffffu), (int)(dm >> 16)));
          const uint32_t lim2 = (uint32_t)min(dmin + min(a.max_dist_thr, 0x3ffe), 0x3ffe) * 0x00010001u;   // (< 0x3fff: dead slots never qualify)
          uint32_t okx[NR];
#pragma unroll
          for (int r = 0; r < NR; ++r) okx[r] = pk_sub(lim2, d2[r]);   // sign clear: d - dmin <= threshold
          const uint32_t okm = mask16(okx);
          const int lc = okm ? (int)lane_ffbl(okm) : 0x7fffffff, hc = okm ? 31 - (int)__builtin_clz(okm) : -0x7fffffff;
          const int akj = jt;   // slot of the end diagonal tlen - plen
          int new_lo = lo, new_hi = hi;
          const int top_limit = min(akj, hi);
          if (top_limit > lo) new_lo = min(lc, top_limit);
          const int bottom_limit = max(akj, new_lo);
          if (bottom_limit < hi) new_hi = max(hc, bottom_limit);
          if (consider) steps_wait = a.steps_between;
          const bool cut = consider && (new_lo != lo || new_hi != hi);
          if (__any(cut)) {
            // the dropped slots read NULL in M, I and D from now on (slots below new_lo or above new_hi: the signs of slot - new_lo and
            // new_hi - slot on packed halves; a pair without a cut keeps all)
            const uint32_t lo2 = (uint32_t)(cut ? new_lo : 0) * 0x00010001u, hi2 = (uint32_t)(cut ? new_hi : W - 1) * 0x00010001u;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              const uint32_t idx2 = (uint32_t)(2 * r) | ((uint32_t)(2 * r + 1) << 16);
              const uint32_t out = pk_sub(idx2, lo2) | pk_sub(hi2, idx2);
              const uint32_t km = ~__builtin_bit_cast(uint32_t, __builtin_bit_cast(lane_s2, out) >> (short)15);   // 0xffff per slot that stays
              cur[r] = (cur[r] & km) | (WFA_LANE_NULL2 & ~km);
              Ih[0][r] = (Ih[0][r] & km) | (WFA_LANE_NULL2 & ~km);
              Dh[0][r] = (Dh[0][r] & km) | (WFA_LANE_NULL2 & ~km);
            }
          }
        }
      }
    }

    // =================== compute-next (R/wavefront_compute_affine.c:44-86) ===================
    {
      uint32_t gi[NR], gd[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const uint32_t mo = (OE == 1) ? cur[r] : Mh[OE - 2][r];   // M at s - (o + e): depth OE counted from the new score
        gi[r] = LIN ? mo : pk_max(mo, Ih[E - 1][r]);
        gd[r] = LIN ? mo : pk_max(mo, Dh[E - 1][r]);
      }
      uint32_t nm[NR], ni[NR], nd[NR];
      const uint32_t one2 = 0x00010001u;
      // FULL: per slot, is the extension of I / D below its opening?  (the sign of I_e - M_oe / D_e - M_oe, read by the neighbour slot)
      uint32_t cmp_i[FULL ? NR : 1], cmp_d[FULL ? NR : 1], cbyte[FULL ? NR : 1];
      if constexpr (FULL) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const uint32_t mo = (OE == 1) ? cur[r] : Mh[OE - 2][r];
          cmp_i[r] = LIN ? 0x80008000u : pk_sub(Ih[E - 1][r], mo);   // (LIN: every gap is an opening)
          cmp_d[r] = LIN ? 0x80008000u : pk_sub(Dh[E - 1][r], mo);
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        // I(k) = max(M_oe, I_e)(k - 1) + 1: the value of slot j - 1; D(k) = max(M_oe, D_e)(k + 1): slot j + 1
        const uint32_t below = __builtin_amdgcn_alignbit(gi[r], (r > 0) ? gi[r - 1] : WFA_LANE_NULL2, 16);
        const uint32_t above = __builtin_amdgcn_alignbit((r < NR - 1) ? gd[r + 1] : WFA_LANE_NULL2, gd[r], 16);
        ni[r] = pk_add(below, one2);
        nd[r] = above;
        const uint32_t mx = (X == 1) ? cur[r] : Mh[X - 2][r];
        const uint32_t gap = pk_max(nd[r], ni[r]), x1 = (LIN == 2) ? WFA_LANE_NULL2 : pk_add(mx, one2);
        nm[r] = pk_clamp(pk_max(gap, x1), lim[r]);
        if constexpr (FULL) {
          // the four comparison bits of each of the two slots (sign bits 15 / 31 of the packed differences): 8: mismatch below the
          // best gap, 4: deletion below insertion, 2: I extension below opening (slot j - 1), 1: D extension below opening (slot j + 1)
          const uint32_t ca = pk_sub(x1, gap), cb = pk_sub(nd[r], ni[r]);
          const uint32_t cc = __builtin_amdgcn_alignbit(cmp_i[r], (r > 0) ? cmp_i[r - 1] : 0u, 16);
          const uint32_t cd = __builtin_amdgcn_alignbit((r < NR - 1) ? cmp_d[r + 1] : 0u, cmp_d[r], 16);
          uint32_t t4 = (cc >> 2 & 0x20002000u) | (cd >> 3 & 0x10001000u);
          t4 = (cb >> 1 & 0x40004000u) | t4;
          t4 = (ca & 0x80008000u) | t4;
          cbyte[r] = ((t4 >> 12) & 0xFu) | ((t4 >> 24) & 0xF0u);
        }
      }
      if constexpr (FULL) {
        // bytes 0..3 = registers 0..3, 4..7 = registers 4..7
        const uint32_t c0 = cbyte[0] | (cbyte[1] << 8) | (cbyte[2] << 16) | (cbyte[3] << 24);
        const uint32_t c1 = cbyte[4] | (cbyte[5] << 8) | (cbyte[6] << 16) | (cbyte[7] << 24);
        // the record of the step being made: one 512-byte store of the wave (idle lanes write along: nobody reads theirs)
        if (gstep + 1 < a.codes_cap) wave_codes[(unsigned)(gstep + 1) * 64u + (unsigned)lane] = make_uint2(c0, c1);
        else codes_full = true;
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int d = DM - 1; d > 0; --d) Mh[d][r] = Mh[d - 1][r];
        Mh[0][r] = cur[r];
#pragma unroll
        for (int d = E - 1; d > 0; --d) { Ih[d][r] = Ih[d - 1][r]; Dh[d][r] = Dh[d - 1][r]; }
        Ih[0][r] = ni[r]; Dh[0][r] = nd[r];
        cur[r] = nm[r];
      }
      if constexpr (HEUR) {
        // a cell of an outermost slot is alive (an offset or a gap value >= 0): the next steps could reach beyond the band
        edge_live = (((nm[0] & ni[0] & nd[0]) & 0x00008000u) == 0u) || (((nm[NR - 1] & ni[NR - 1] & nd[NR - 1]) & 0x80000000u) == 0u);
      }
    }
    ++gstep;
  }
  if (WFA_LANE_DEBUG_COUNTERS && !FULL && a.hist) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.hist), (unsigned long long)gstep); }
}

#ifndef __HIPCC_RTC__   // ---- host side (launch code) ----
// per-shape entry points (csrc/k_lane.hip compiled once per shape index of WFA_SEG_SHAPES)
#define WFA_LANE_DECL(i, x, oe, e) \
  int launch_lane_s##i(unsigned grid, size_t smem, hipStream_t stream, const FastArgs& a, int slot_words, int refill_min, bool full, int heur);
// (the shape list is wfa_seg.hpp's; declared here without including it)
WFA_LANE_DECL(0, 2, 4, 1) WFA_LANE_DECL(1, 2, 3, 1) WFA_LANE_DECL(2, 4, 7, 1) WFA_LANE_DECL(3, 3, 5, 1)
WFA_LANE_DECL(4, 6, 8, 3) WFA_LANE_DECL(5, 5, 3, 3) WFA_LANE_DECL(6, 1, 2, 1)
#undef WFA_LANE_DECL

template <int X, int OE, int E>
inline int launch_lane_shape(unsigned grid, size_t smem, hipStream_t stream, const FastArgs& a, int slot_words, int refill_min, bool full, int heur) {
  if (heur == 2) hipLaunchKernelGGL((wfa_lane_kernel<X, OE, E, false, true, 0, 16>), dim3(grid), dim3(64), smem, stream, a, slot_words, refill_min);
  else if (heur) hipLaunchKernelGGL((wfa_lane_kernel<X, OE, E, false, true>), dim3(grid), dim3(64), smem, stream, a, slot_words, refill_min);
  else if (full) hipLaunchKernelGGL((wfa_lane_kernel<X, OE, E, true>), dim3(grid), dim3(64), smem, stream, a, slot_words, refill_min);
  else hipLaunchKernelGGL((wfa_lane_kernel<X, OE, E, false>), dim3(grid), dim3(64), smem, stream, a, slot_words, refill_min);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// configurations of the general score-only form (HEUR): gap-affine with an instantiated shape, match 0, no wildcard; no heuristic or
// wf-adaptive (X-drop stays in the banded kernel), any free ends, any step limit (the shape list is wfa_seg.hpp's: seg_shape())
inline bool lane_heur_config(const WfaDevConfig& c, int ncomp) {
  return ncomp == 3 && c.match == 0 && c.wildcard < 0 && (c.heuristic == 0 || c.heuristic == 1);
}

// LDS words of a pair's slot for reads up to max_len bases: both sequences + one spare word, odd (lane slots then
// fall into different banks)
inline int lane_slot_words(int max_len) {
  const int w = 2 * ((max_len + 15) >> 4) + 1;
  return w | 1;
}

// shape_idx: index in WFA_SEG_SHAPES (seg_shape()); per_cu: slices of the work list (waves) per CU
// records of origin codes per lane of the FULL form for a penalty shape (LaneFull<OE, E>::NREC)
inline int lane_full_records(int OE, int E) { return 2 * (OE - E) + E * 17 + 1; }
// FULL form: waves of a launch over `nwork` pairs and the records a wave's list needs.  A wave of `per` pairs makes at most
// per x NREC lane-steps with >= 64 - refill_min - (a burst) lanes busy: per x NREC / 32 + 2 NREC records are never reached in
// practice, and a wave that does fill its list hands the rest of its pairs on.
inline void lane_full_geometry(uint32_t nwork, int cu_count, int per_cu, int min_pairs, int OE, int E, long long* grid_out, int* cap_out) {
  if (min_pairs <= 0) min_pairs = 256;
  long long grid = (long long)cu_count * per_cu;
  const long long max_grid = ((long long)nwork + min_pairs - 1) / std::max(min_pairs, 64);
  if (grid > max_grid) grid = max_grid;
  if (grid < 1) grid = 1;
  const long long per = ((long long)nwork + grid - 1) / grid;
  const int nrec = lane_full_records(OE, E);
  *grid_out = grid;
  *cap_out = (int)(per * nrec / 32 + 2 * nrec + 8);
}

// run records per pair of the FULL form (ints of a slot): an alignment inside the band has at most Bmin / g edits
#define WFA_LANE_RUN_SLOT 32

// full = the FULL form: a.hist = run-record slots (a.hist_stride ints each, slot = work item - a.work_begin), a.end_state per slot
// (shape_idx WFA_SHAPE_RTC: no instantiation in the library — the kernel of (X, OE, E) is compiled at run time, csrc/wfa_rtc.cpp)
inline int launch_lane_args(int shape_idx, int OE, int E, int cu_count, int per_cu, int refill_min, int max_len, hipStream_t stream, FastArgs a, bool full, int lds_pad_kb = 0, int min_pairs = 0, int heur = 0, int X = 0);   // heur: 0 no, 1 the general form (16 diagonals), 2 its 32-diagonal form

inline int launch_lane(int shape_idx, int g, int cu_count, int per_cu, int refill_min, int max_len, hipStream_t stream, const uint32_t* words,
                       const WfaPairMeta* meta, const uint32_t* worklist, const uint32_t* nwork_dev, uint32_t nwork,
                       int32_t* score, int32_t* status, uint32_t* fb_list, uint32_t* fb_count, int32_t* debug_counters = nullptr, int lds_pad_kb = 0,
                       int X = 0, int OE = 0, int E = 0, int min_pairs = 0, uint32_t* dyn_next = nullptr, uint32_t dyn_chunk = 0) {
  FastArgs a = FastArgs();
  a.dyn_next = dyn_chunk ? dyn_next : nullptr; a.dyn_chunk = dyn_chunk;
  a.words = words; a.meta = meta; a.worklist = worklist; a.nwork_dev = nwork_dev; a.nwork = nwork;
  a.score = score; a.status = status; a.fb_list = fb_list; a.fb_count = fb_count;
  a.g = g;
  a.hist = debug_counters; a.hist_stride = 0; a.end_state = nullptr; a.work_begin = 0;
  a.ef = a.pbf = a.pef = a.tbf = a.tef = 0; a.heur = 0; a.min_wf_len = a.max_dist_thr = a.steps_between = 0; a.max_steps = INT_MAX;
  return launch_lane_args(shape_idx, OE, E, cu_count, per_cu, refill_min, max_len, stream, a, false, lds_pad_kb, min_pairs, 0, X);
}

inline int launch_lane_args(int shape_idx, int OE, int E, int cu_count, int per_cu, int refill_min, int max_len, hipStream_t stream, FastArgs a, bool full, int lds_pad_kb, int min_pairs, int heur, int X) {
  const uint32_t nwork = a.nwork;
  const uint32_t* nwork_dev = a.nwork_dev;
  const int slot_words = lane_slot_words(std::min(max_len, WFA_FAST_MAX_LEN));
  const size_t smem = ((size_t)64 * slot_words + 8) * sizeof(uint32_t) + (size_t)lds_pad_kb * 1024;   // (lds_pad_kb: occupancy experiments)   // (lds_pad_kb: occupancy experiments)
  // every wave should see several hundred pairs (64 lanes x a few refills), and there should be several waves per SIMD
  long long grid = (long long)cu_count * per_cu;
  // a wave should see a few refills' worth of pairs (256) — when the batch is large enough to give every SIMD four such waves.  A smaller
  // batch is cut into more waves of fewer pairs, down to one lane-full each: a run is then as long as a wave's life, and 256 pairs per
  // wave made every score-only run of up to 260 k pairs last 275 us (65 536 pairs: 274 -> 155 us with 64; 1 M pairs: 550 us with 256, 616
  // with 64; tools/probes/midbatch_probe.py).  The full-CIGAR form keeps 256 (its walks like long lists: C1 1.06 ms against 1.11).
  if (min_pairs <= 0) min_pairs = full ? 256 : (int)std::max<long long>(64, std::min<long long>(256, (long long)nwork / ((long long)cu_count * 16)));
  const long long max_grid = ((long long)nwork + min_pairs - 1) / std::max(min_pairs, 64);
  if (!nwork_dev && grid > max_grid) grid = max_grid;
  if (grid < 1) grid = 1;
  if (full && a.codes_cap <= 0) return -1;   // (the caller sizes the record lists with lane_full_geometry)
  if (a.lin) shape_idx = WFA_SHAPE_RTC;      // (the one-component form: instantiated at run time whatever the shape)
  switch (shape_idx) {
    case 0: return launch_lane_s0((unsigned)grid, smem, stream, a, slot_words, refill_min, full, heur);
    case 1: return launch_lane_s1((unsigned)grid, smem, stream, a, slot_words, refill_min, full, heur);
    case 2: return launch_lane_s2((unsigned)grid, smem, stream, a, slot_words, refill_min, full, heur);
    case 3: return launch_lane_s3((unsigned)grid, smem, stream, a, slot_words, refill_min, full, heur);
    case 4: return launch_lane_s4((unsigned)grid, smem, stream, a, slot_words, refill_min, full, heur);
    case 5: return launch_lane_s5((unsigned)grid, smem, stream, a, slot_words, refill_min, full, heur);
    case 6: return launch_lane_s6((unsigned)grid, smem, stream, a, slot_words, refill_min, full, heur);
    case WFA_SHAPE_RTC: {
      struct { FastArgs a; int slot_words; int refill_min; } args = {a, slot_words, refill_min};   // (the kernel's argument list)
      const std::string name = "wfa::wfa_lane_kernel<" + std::to_string(X) + ", " + std::to_string(OE) + ", " + std::to_string(E) + ", " +
                               rtc_bool(full && !heur) + ", " + rtc_bool(heur != 0) + (heur == 2 ? std::string(", 0, 16>") : a.lin ? ", " + std::to_string(a.lin) + ">" : std::string(">"));
      return rtc_launch("wfa_lane.hpp", name, (unsigned)grid, 64, smem, stream, &args, sizeof(args));
    }
    default: return -1;
  }
}

#endif  // __HIPCC_RTC__

}  // namespace wfa
