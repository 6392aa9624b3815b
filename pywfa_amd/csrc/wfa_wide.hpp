// wfa_wide.hpp — exact gap-affine alignment of long reads whose wavefronts outgrow the 256-diagonal register window of
// wfa_band.hpp: ONE alignment per workgroup, the M / I / D wavefronts of the last max(x, o+e)/g + 1 score steps in LDS
// as rows of int16 offsets indexed by diagonal (north_star's layout), the two packed sequences in
// LDS beside them.  10 kb reads at 8 % without a heuristic: wavefronts grow to ~4 500 diagonals, 9 rows x 5 000
// diagonals x 2 B + 5 KB of sequences = 95 KB of the CU's 160 KB.
//
// What a step does (R = /root/reference/pywfa/WFA2_lib/wavefront), for score s = t g (only multiples of g = gcd(x, o+e, e)
// are reachable; the scores in between are the reference's null steps):
//   compute-next  R/wavefront_compute_affine.c:44-86 (I = max(M[s-o-e], I[s-e])(k-1) + 1, D = max(M[s-o-e], D[s-e])(k+1),
//                 M = max(M[s-x] + 1, I, D), only M clamped to the sequences), range by R/wavefront_compute.c:40-86,
//                 ends trimmed to the first / last in-bounds cell per component (R/wavefront_compute.c:571-605)
//   extend        R/wavefront_extend_kernels.c:64-88 on the 2-bit codes, 32 bases per round, fused into the same pass
//   termination   R/wavefront_termination.c:37-61 (end-to-end), :115-162 (ends-free, lowest k wins)
//   limit         R/wavefront_unialign.c:102-107 (max_steps)
// Two barriers per step: after the pass (rows + trimmed limits visible), after the few end cells of I / D outside the
// trimmed limits were set to NULL.  Invariant: every row is NULL outside its trimmed [lo, hi], so the pass reads its five
// inputs without range tests.  Offsets are < 32 768 (plen + tlen <= 32 000), NULL = -16 384 in a row; a negative value
// is dead (it never becomes in-bounds, R/wavefront_offset.h:44-57).
//
// Scope: gap-affine and gap-affine-2p, match = 0, 2-bit pairs, no heuristic, end-to-end or ends-free, score-only or full
// CIGAR.  Full CIGAR keeps the piggy-back history of the general kernel (one byte of origin codes per cell + a 12-byte
// directory record per step in the workgroup's slice of the HBM workspace; wfa_general.hpp PB) and the same walk / forward
// unpack.  A pair whose wavefront leaves the rows, whose history does not fit, or that runs into an all-NULL stretch is
// handed on (fb_list) to the next stage.
//
// Two forms of the rows (template GROWS): in LDS (gap-affine: as above, ~7 900 diagonals at most), or in the workgroup's slice
// of the HBM workspace, as wide as the whole diagonal range of the longest pair — what outgrew the LDS rows, and gap-affine-2p
// (TWO: components M, I1, D1, I2, D2, R/wavefront_compute_affine2p.c:45-106; the M ring alone is max(x, o+e, o2+e2)/g + 1
// rows: 26 for pywfa's defaults, 37 rows x 20 000 diagonals x 2 B = 1.5 MB per workgroup at 10 kb): BASELINE's C4 as written.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <type_traits>
#include "wfa_common.hpp"
#include "wfa_hip.h"

namespace wfa {

struct WideArgs {
  const uint32_t* words;
  const WfaPairMeta* meta;
  const uint32_t* worklist;   // nullptr = identity
  const uint32_t* nwork_dev;  // non-null: count read from device memory
  uint32_t nwork;
  int32_t* score;
  int32_t* status;
  uint32_t* fb_list;
  uint32_t* fb_count;
  uint8_t* cigar_ops;
  const int64_t* cigar_off;
  int64_t* cigar_begin;
  int32_t* cigar_len;
  int32_t* hist;          // full scope: slice of workgroup b = hist + b * hist_stride (ints)
  long long hist_stride;
  short* rows;            // rows in the HBM workspace: slice of workgroup b = rows + b * rows_stride (elements: int16, or int32 in the W32 form)
  long long rows_stride;
  int g, X, OE, E;        // score step and the penalties in steps
  int OE2, E2;            // gap-affine-2p: the second gap piece
  int ef, pbf, pef, tbf, tef;
  int max_steps;
  int wcap;               // diagonals per row
  int seq_words;          // LDS words per sequence (>= words of the longest sequence + 3)
  int heur, min_wf_len, max_dist_thr, steps_between;   // heur = 1: wf-adaptive (R/wavefront_heuristic.c:257-293), what the banded stages handed on
};

#define WFA_WIDE_NULL (-16384)
#define WFA_WIDE_CTRL_INTS 64

__device__ __forceinline__ uint32_t wide_ffbl(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// minimum over the wave without LDS traffic (the step of a lone alignment is a dependent chain: six ds_bpermute hops per reduction
// were a fifth of it): butterfly inside each row of 16 lanes with DPP, then the four row leaders
__device__ __forceinline__ int wide_wave_min(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wide_wave_max(int v) { return ~wide_wave_min(~v); }

// rows of one alignment: M for the last max(x, o+e, o2+e2)/g + 1 steps, I1 / D1 for the last e/g + 1, I2 / D2 for the last
// e2/g + 1, and one row that is always NULL (inputs before score 0)
static inline int wide_rows(int X, int OE, int E, int OE2, int E2) {
  int dm = X > OE ? X : OE;
  if (OE2 > dm) dm = OE2;
  return dm + 1 + 2 * (E + 1) + (OE2 > 0 ? 2 * (E2 + 1) : 0) + 1;
}
static inline size_t wide_row_halfs(int wcap) { return (size_t)((wcap + 2 + 1) & ~1); }
// LDS of a workgroup: control words, row limits, the two sequences, and (rows in LDS) the rows
static inline size_t wide_smem_bytes(int X, int OE, int E, int OE2, int E2, int wcap, int seq_words, bool rows_in_lds, int offset_bytes = 2) {
  const int NR = wide_rows(X, OE, E, OE2, E2);
  return (size_t)(WFA_WIDE_CTRL_INTS + 2 * NR) * 4 + (size_t)2 * seq_words * 4 + (rows_in_lds ? (size_t)NR * wide_row_halfs(wcap) * (size_t)offset_bytes : 0);
}

// Full CIGAR of the wide-wavefront kernels (this file and wfa_tile.hpp): one lane walks the origin codes back from the end cell
// (R/wavefront_backtrace.c:320-529's choices, decided at compute time), keeping one event byte per edit, then unpacks forwards
// from the start cell re-extending the matches on the LDS copies of the packed sequences.  Directory record of step t:
// hist[hist_stride - 3 (t + 1)] = {lo, hi, first code byte}.  Returns the number of ops written to `out`, or -1 when the
// events do not fit `ev_cap` or the walk leaves score 0 below.
template <bool TWO>
__device__ inline long long wide_walk_unpack(const int* hist, long long hist_stride, const uint8_t* pb_codes, uint8_t* ev, long long ev_cap,
                                             int end_t, int end_k, int X, int OE, int E, int OE2, int E2,
                                             const uint32_t* sP, const uint32_t* sT, int plen, int tlen, uint8_t* out) {
  int tc = end_t, k = end_k, comp = 0;
  long long nev = 0;
  while (tc > 0) {
    if (nev >= ev_cap) return -1;
    const int* d = hist + hist_stride - 3ll * (tc + 1);
    const int cd = (k >= d[0] && k <= d[1]) ? pb_codes[(long long)d[2] + (k - d[0])] : 0;
    const uint8_t flag = (comp == 0) ? 0x80 : 0;
    int src;   // 0 mismatch, 1 D1, 2 D2, 3 I1, 4 I2
    if (TWO) src = (comp == 0) ? (cd & 7) : (comp == 1) ? 3 : (comp == 2) ? 1 : (comp == 3) ? 4 : 2;
    else src = (comp == 0) ? ((cd & 3) == 0 ? 0 : ((cd & 3) == 1 ? 1 : 3)) : (comp == 1 ? 3 : 1);
    const int bi1 = TWO ? 8 : 4, bd1 = TWO ? 16 : 8;
    if (src == 0) { ev[nev++] = (uint8_t)('X' | 0x80); tc -= X; }
    else if (src == 1) { ev[nev++] = (uint8_t)('D' | flag); ++k; if (cd & bd1) { tc -= E; comp = 2; } else { tc -= OE; comp = 0; } }
    else if (src == 2) { ev[nev++] = (uint8_t)('D' | flag); ++k; if (cd & 64) { tc -= E2; comp = 4; } else { tc -= OE2; comp = 0; } }
    else if (src == 3) { ev[nev++] = (uint8_t)('I' | flag); --k; if (cd & bi1) { tc -= E; comp = 1; } else { tc -= OE; comp = 0; } }
    else { ev[nev++] = (uint8_t)('I' | flag); --k; if (cd & 32) { tc -= E2; comp = 3; } else { tc -= OE2; comp = 0; } }
  }
  if (tc < 0) return -1;
  long long n = 0;
  auto emit = [&](char c, int cnt) { for (int i = 0; i < cnt; ++i) out[n++] = (uint8_t)c; };
  auto lcp = [&](int v, int h) {   // common prefix of pattern[v..] and text[h..] on the LDS copies
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
  return n;
}

// FULL: piggy-back history + walk; TWO: gap-affine-2p (components M, I1, D1, I2, D2); GROWS: the rows live in the workgroup's
// slice of the HBM workspace (L2-resident) instead of LDS — the 2p form: 37 rows x 20 000 diagonals for 10 kb reads
// W32: rows of int32 offsets: reads beyond 16 kb (plen + tlen > 32 000), any number of steps.  (Round 4: also with the rows in LDS —
// the wf-adaptive leftovers of the banded stages, a few hundred diagonals wide: a step without a round trip to L2)
template <bool FULL, bool TWO, bool GROWS, bool W32 = false>
__global__ void __launch_bounds__(1024)
wfa_wide_kernel(const WideArgs a) {
  typedef typename std::conditional<W32, int, short>::type row_t;
  constexpr int RNULL = W32 ? WFA_OFFSET_NULL : WFA_WIDE_NULL;
  constexpr int NC = TWO ? 5 : 3;
  extern __shared__ int wsm[];
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
  const int DM = TWO ? max(max(a.X, a.OE), a.OE2) : max(a.X, a.OE);
  const int NM = DM + 1, NG1 = a.E + 1, NG2 = TWO ? a.E2 + 1 : 0, NR = NM + 2 * NG1 + 2 * NG2 + 1;
  const int rw = (a.wcap + 2 + 1) & ~1;            // halfs per row: guard, wcap diagonals, guard (+ pad)
  int* const ctrl = wsm;                           // three scratch slots of 16 ints (layout at the step loop)
  int* const rlo = wsm + WFA_WIDE_CTRL_INTS;       // trimmed limits of every row
  int* const rhi = rlo + NR;
  uint32_t* const sP = reinterpret_cast<uint32_t*>(rhi + NR);
  uint32_t* const sT = sP + a.seq_words;
  row_t* rows;
  if constexpr (GROWS) rows = reinterpret_cast<row_t*>(a.rows) + (long long)blockIdx.x * a.rows_stride;   // (rows_stride in elements)
  else rows = reinterpret_cast<row_t*>(sT + a.seq_words);
  const int NULLROW = NR - 1;
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  int* const hist = FULL ? a.hist + (long long)blockIdx.x * a.hist_stride : nullptr;

  for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    const WfaPairMeta pm = a.meta[pair];
    const int plen = pm.plen, tlen = pm.tlen;
    const int ak = tlen - plen;
    const int pbf = a.ef ? a.pbf : 0, tbf = a.ef ? a.tbf : 0;
    // rows are centred between the start and the target diagonals
    const int koff = a.wcap / 2 + 1 - (ak + tbf - pbf) / 2;     // row index of diagonal k = k + koff (1 .. wcap)
    const int kmin = 1 - koff, kmax = a.wcap - koff;
    __syncthreads();   // the previous pair is done with LDS and the rows
    // ---- sequences and rows ----
    {
      const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
      const uint32_t* gp = a.words + pm.p_woff;
      const uint32_t* gt = a.words + pm.t_woff;
      for (int i = tid; i < a.seq_words; i += T) { sP[i] = (i < nwp) ? gp[i] : 0u; sT[i] = (i < nwt) ? gt[i] : 0u; }
      // (of the rows only the guard cell below kmin needs a value — NULL: every read goes through the limits of its row)
      for (int i = tid; i < NR; i += T) { rlo[i] = 1; rhi[i] = -1; rows[(long long)i * rw] = (row_t)RNULL; }
      if (tid < 48) { const int j = tid & 15; ctrl[tid] = (j < 5 || (j >= 10 && j <= 12)) ? INT_MAX : INT_MIN; }   // three scratch slots (layout at the step loop)
    }
    bool hand_on = (!W32 && plen + tlen > 32000) || (-pbf < kmin) || (tbf > kmax) || (ak < kmin) || (ak > kmax);
    int end_reason = 0;   // 1 reached, 3 handed on, 4 step limit
    int end_k = 0, end_t = 0;
    long long pb_used = 0;                                  // FULL: code bytes in use
    uint8_t* const pb_codes = FULL ? reinterpret_cast<uint8_t*>(hist) : nullptr;
    const long long pb_cap = FULL ? a.hist_stride * 4 : 0;  // bytes shared by codes (bottom-up) and directory (top-down)
    int null_run = 0;
    int steps_wait = a.steps_between;   // wf-adaptive: steps until the cut-off is looked at again
    __syncthreads();

    int tM = 0, tG1 = 0, tG2 = 0, tS = 0;   // t mod NM, NG1, NG2, 3 (carried from step to step: no division by run-time values in a step)
    for (int t = 0; !hand_on; ++t, tM = (tM + 1 == NM) ? 0 : tM + 1, tG1 = (tG1 + 1 == NG1) ? 0 : tG1 + 1, tG2 = (TWO && tG2 + 1 < NG2) ? tG2 + 1 : 0,
             tS = (tS == 2) ? 0 : tS + 1) {
      const int s = t * a.g;
      // scratch of this step: one of three slots (the slot of step t + 2 is reset during step t, behind this step's barrier: no barrier
      // of its own) — [0 .. NC) trim min, [5 .. 5 + NC) trim max, [10] end k, [11] smallest distance, [12] / [13] first / last kept diagonal
      int* const TRmin = ctrl + 16 * tS;
      int* const TRmax = TRmin + 5;
      // the limit is tested after compute-next of a score and before its extension (R/wavefront_unialign.c:98-107)
      if (t > 0 && s >= a.max_steps) { end_reason = 4; break; }
      if (!W32 && t > 16000) { end_reason = 3; break; }   // (int16 rows: a NULL gains at most 1 per step and must stay negative)
      // ---- rows of this step (M, I1, D1, I2, D2) and its inputs ----
      int rW[NC];
      rW[0] = tM; rW[1] = NM + tG1; rW[2] = NM + NG1 + tG1;
      if (TWO) { rW[3] = NM + 2 * NG1 + tG2; rW[4] = NM + 2 * NG1 + NG2 + tG2; }
      auto back = [](int pos, int lag, int n) { const int x = pos - lag; return x < 0 ? x + n : x; };   // (lag < n)
      const int iX = (t >= a.X) ? back(tM, a.X, NM) : NULLROW;
      const int iO = (t >= a.OE) ? back(tM, a.OE, NM) : NULLROW;
      const int iI = (t >= a.E) ? NM + back(tG1, a.E, NG1) : NULLROW;
      const int iD = (t >= a.E) ? NM + NG1 + back(tG1, a.E, NG1) : NULLROW;
      const int iO2 = (TWO && t >= a.OE2) ? back(tM, a.OE2, NM) : NULLROW;
      const int iI2 = (TWO && t >= a.E2) ? NM + 2 * NG1 + back(tG2, a.E2, NG2) : NULLROW;
      const int iD2 = (TWO && t >= a.E2) ? NM + 2 * NG1 + NG2 + back(tG2, a.E2, NG2) : NULLROW;
      // Trimmed limits of the input rows.  A row is READ through them (a cell outside reads NULL): what a row still holds of the wavefront
      // it held NM steps ago, its cells the trimming or the cut-off dropped, and a pair's first steps need no store of NULLs at all —
      // those loops (up to 17 of them, each a divergent loop) were most of a narrow wavefront's step.
      const int xl = rlo[iX], xh = rhi[iX], ol = rlo[iO], oh = rhi[iO], il = rlo[iI], ih = rhi[iI], dl = rlo[iD], dh = rhi[iD];
      int o2l = 1, o2h = -1, i2l = 1, i2h = -1, d2l = 1, d2h = -1;
      if (TWO) { o2l = rlo[iO2]; o2h = rhi[iO2]; i2l = rlo[iI2]; i2h = rhi[iI2]; d2l = rlo[iD2]; d2h = rhi[iD2]; }
      int lo, hi;
      if (t == 0) { lo = -pbf; hi = tbf; }
      else {
        // R/wavefront_compute.c:40-86 (a null input counts with lo = 1, hi = -1, as there)
        lo = min(min(xl, ol - 1), min(il + 1, dl - 1));
        hi = max(max(xh, oh + 1), max(ih + 1, dh - 1));
        bool all_null = xl > xh && ol > oh && il > ih && dl > dh;
        if (TWO) {
          lo = min(lo, min(o2l - 1, min(i2l + 1, d2l - 1)));
          hi = max(hi, max(o2h + 1, max(i2h + 1, d2h - 1)));
          all_null = all_null && o2l > o2h && i2l > i2h && d2l > d2h;
        }
        if (all_null) { lo = 1; hi = -1; }
      }
      if (lo <= hi && (lo < kmin || hi > kmax)) { end_reason = 3; break; }
      if (lo > hi) { if (++null_run > DM + 2) { end_reason = 3; break; } } else null_run = 0;
      long long code_base = 0;
      if (FULL) {
        const long long nb = (lo <= hi) ? (long long)hi - lo + 1 : 0;
        if (pb_used + nb + (long long)(t + 2) * 12 + 64 > pb_cap || pb_used + nb > 0x7fffff00ll) { end_reason = 3; break; }
        code_base = pb_used; pb_used += nb;
        if (tid == 0) { int* d = hist + a.hist_stride - 3ll * (t + 1); d[0] = (lo <= hi) ? lo : 1; d[1] = (lo <= hi) ? hi : 0; d[2] = (int)code_base; }
      }
      row_t* wR[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) wR[c] = rows + (long long)rW[c] * rw + koff;
      const row_t* const pX = rows + (long long)iX * rw + koff;
      const row_t* const pO = rows + (long long)iO * rw + koff;
      const row_t* const pI = rows + (long long)iI * rw + koff;
      const row_t* const pD = rows + (long long)iD * rw + koff;
      const row_t* const pO2 = rows + (long long)iO2 * rw + koff;
      const row_t* const pI2 = rows + (long long)iI2 * rw + koff;
      const row_t* const pD2 = rows + (long long)iD2 * rw + koff;
      // (the index is redirected to the row's guard cell, which holds NULL, rather than the value selected afterwards: a load whose value is
      // used under a condition is moved under it by the compiler — five loads, each behind its own branch and wait)
      const int kg = kmin - 1;
      auto rd = [kg](const row_t* p, int k, int l, int h) -> int { return p[((unsigned)(k - l) <= (unsigned)(h - l) && l <= h) ? k : kg]; };
      // ---- the pass: compute, clamp, extend, store; trimmed limits by wave ballots ----
      int wmin[NC], wmax[NC];   // (wave-uniform)
#pragma unroll
      for (int c = 0; c < NC; ++c) { wmin[c] = INT_MAX; wmax[c] = INT_MIN; }
      // wf-adaptive: the smallest distance to the end over the extended M cells, and this thread's cell if it had just one
      int dloc = INT_MAX, my_cells = 0, my_off = RNULL, my_k = 0;
      // diagonals whose every input lies inside its row's limits: a wave whose 64 diagonals are all of that kind reads without the tests
      // (a wide exact wavefront is mostly such waves)
      int fl = max(max(xl, ol + 1), max(il + 1, dl - 1)), fh = min(min(xh, oh - 1), min(ih + 1, dh - 1));
      if (TWO) { fl = max(fl, max(o2l + 1, max(i2l + 1, d2l - 1))); fh = min(fh, min(o2h - 1, min(i2h + 1, d2h - 1))); }
      for (int k0 = lo + (tid & ~63); k0 <= hi; k0 += T) {
        const int k = k0 + lane;
        const bool in = k <= hi;
        const bool inner = k0 >= fl && k0 + 63 <= fh;   // (wave-uniform)
        int v5[NC];   // M, I1, D1, I2, D2 of this diagonal
#pragma unroll
        for (int c = 0; c < NC; ++c) v5[c] = RNULL;
        int code = 0;
        if (in) {
          if (t == 0) {
            v5[0] = max(k, 0);   // R/wavefront_aligner.c:251-310: offset 0 on diagonal 0, the free begins on theirs
          } else {
            int mo_lo, ie_lo, mo_hi, de_hi, xv;
            if (inner) { mo_lo = pO[k - 1]; ie_lo = pI[k - 1]; mo_hi = pO[k + 1]; de_hi = pD[k + 1]; xv = pX[k]; }
            else { mo_lo = rd(pO, k - 1, ol, oh); ie_lo = rd(pI, k - 1, il, ih); mo_hi = rd(pO, k + 1, ol, oh); de_hi = rd(pD, k + 1, dl, dh); xv = rd(pX, k, xl, xh); }
            v5[1] = max(mo_lo, ie_lo) + 1;
            v5[2] = max(mo_hi, de_hi);
            const int x1 = xv + 1;
            if (TWO) {
              // R/wavefront_compute_affine2p.c:45-106
              int mo2_lo, i2e_lo, mo2_hi, d2e_hi;
              if (inner) { mo2_lo = pO2[k - 1]; i2e_lo = pI2[k - 1]; mo2_hi = pO2[k + 1]; d2e_hi = pD2[k + 1]; }
              else { mo2_lo = rd(pO2, k - 1, o2l, o2h); i2e_lo = rd(pI2, k - 1, i2l, i2h); mo2_hi = rd(pO2, k + 1, o2l, o2h); d2e_hi = rd(pD2, k + 1, d2l, d2h); }
              v5[3] = max(mo2_lo, i2e_lo) + 1;
              v5[4] = max(mo2_hi, d2e_hi);
              const int best = max(max(v5[2], v5[4]), max(x1, max(v5[1], v5[3])));
              v5[0] = best;
              if (FULL) {
                // the backtrace's choice on equal offsets (R/wavefront_backtrace.c:49-59): mismatch > D2 > D1 > I2 > I1,
                // extension > opening (as wfa_general.hpp PB)
                const int mc = (x1 >= best) ? 0 : (v5[4] >= best) ? 2 : (v5[2] >= best) ? 1 : (v5[3] >= best) ? 4 : 3;
                code = mc | ((ie_lo >= mo_lo) ? 8 : 0) | ((de_hi >= mo_hi) ? 16 : 0) | ((i2e_lo >= mo2_lo) ? 32 : 0) | ((d2e_hi >= mo2_hi) ? 64 : 0);
              }
            } else {
              v5[0] = max(v5[2], max(x1, v5[1]));
              if (FULL) {
                const int mc = (x1 >= max(v5[2], v5[1])) ? 0 : ((v5[2] >= v5[1]) ? 1 : 2);
                code = mc | ((ie_lo >= mo_lo) ? 4 : 0) | ((de_hi >= mo_hi) ? 8 : 0);
              }
            }
          }
        }
        // in bounds: max(k, 0) <= offset <= min(tlen, plen + k) (0 <= h <= tlen and 0 <= v <= plen)
        const int limk = min(tlen, plen + k);   // the largest offset on diagonal k
        const int base = max(k, 0);
        const bool kin = in && limk >= base;    // (a diagonal beyond -plen .. tlen holds no cell)
        const uint32_t span = (uint32_t)(limk - base);
        bool inb[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) inb[c] = (uint32_t)(v5[c] - base) <= span && kin;
        const bool m_in = inb[0];
        if (!m_in) v5[0] = RNULL;               // only M is clamped (R/wavefront_compute_affine.c:80-84)
        unsigned long long bm = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const unsigned long long bc = __ballot(inb[c]);
          if (c == 0) bm = bc;
          if (bc) { wmin[c] = min(wmin[c], k0 + (int)__builtin_ctzll(bc)); wmax[c] = max(wmax[c], k0 + 63 - (int)__builtin_clzll(bc)); }
        }
        // extend M (R/wavefront_extend_kernels.c:64-88): 32 bases per round, never past either sequence end
        if (bm) {
          int h = v5[0], v = v5[0] - k, left = m_in ? limk - v5[0] : 0;
          bool more = false;
          if (left > 0) {
            // first probe: 16 bases (cells away from the alignment path compare unrelated bases and stop at once)
            const int pi = v >> 4, ti = h >> 4;
            const uint32_t x = __builtin_amdgcn_alignbit(sP[pi + 1], sP[pi], (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(sT[ti + 1], sT[ti], (uint32_t)h << 1);
            const int m = min((int)(wide_ffbl(x) >> 1), min(16, left));
            v += m; h += m; left -= m;
            more = (m == 16) && (left > 0);
          }
          while (__any(more)) {
            if (more) {
              const int pi = v >> 4, ti = h >> 4;
              const uint32_t p0 = sP[pi], p1 = sP[pi + 1], p2 = sP[pi + 2], t0 = sT[ti], t1 = sT[ti + 1], t2 = sT[ti + 2];
              const uint32_t xl_ = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
              const uint32_t xh_ = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
              const uint32_t fb = min(wide_ffbl(xl_), wide_ffbl(xh_) | 32u);   // (v_ffbl_b32 gives ~0 for 0)
              const int m = min((int)(fb >> 1), min(32, left));
              v += m; h += m; left -= m;
              more = (m == 32) && (left > 0);
            }
          }
          if (m_in) {
            v5[0] = h;
            // termination on the extended offset
            if (a.ef) {
              if ((h >= tlen && plen - v <= a.pef) || (v >= plen && tlen - h <= a.tef)) atomicMin(&TRmin[10], k);
            } else if (k == ak && h >= tlen) {
              TRmin[10] = k;
            }
          }
        }
        if (a.heur == 1 && in) {
          if (v5[0] >= 0) dloc = min(dloc, max(tlen, plen + k) - v5[0]);   // max(plen - v, tlen - h)
          my_off = v5[0]; my_k = k; ++my_cells;
        }
        if (in) {
          // (negative gap values are stored as they are: they start at NULL and gain at most 1 per step, so they stay
          // negative for the 16 000 steps a pair may take here, and a negative offset is never in bounds)
#pragma unroll
          for (int c = 0; c < NC; ++c) wR[c][k] = (row_t)v5[c];
          if (FULL) pb_codes[code_base + (k - lo)] = (uint8_t)code;
        }
      }
      if (a.heur == 1) dloc = wide_wave_min(dloc);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) if (wmin[c] != INT_MAX) { atomicMin(&TRmin[c], wmin[c]); atomicMax(&TRmax[c], wmax[c]); }
        if (a.heur == 1 && dloc != INT_MAX) atomicMin(&TRmin[11], dloc);
      }
      __syncthreads();   // rows, trimmed limits, the end flag and the smallest distance of this step are visible
      // ---- trimmed limits (R/wavefront_compute.c:571-605): first / last in-bounds cell; none -> null ----
      int tlo[NC], thi[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int mn = TRmin[c], mx = TRmax[c];
        if (mn != INT_MAX) { tlo[c] = mn; thi[c] = mx; } else { tlo[c] = 1; thi[c] = -1; }
      }
      const int ek = TRmin[10];
      if (ek != INT_MAX) { end_reason = 1; end_k = ek; end_t = t; }
      if (tid == 0) {   // the slot of step t + 2 (last read in step t - 1; next written behind the barrier of step t + 1)
        int* o = ctrl + 16 * ((tS == 0) ? 2 : tS - 1);
#pragma unroll
        for (int c = 0; c < 5; ++c) { o[c] = INT_MAX; o[5 + c] = INT_MIN; }
        o[10] = INT_MAX; o[11] = INT_MAX; o[12] = INT_MAX; o[13] = INT_MIN;
      }
      // ---- wf-adaptive cut-off on the extended M wavefront (R/wavefront_heuristic.c:257-293, dispatcher :509-567): the diagonals whose
      // distance to the end exceeds the smallest by more than the threshold are dropped from both ends, never past the end diagonal;
      // the gap wavefronts are cut to the same limits (the equate) ----
      if (a.heur == 1 && !end_reason && tlo[0] <= thi[0]) {
        --steps_wait;
        const int mlo = tlo[0], mhi = thi[0];
        if (steps_wait <= 0 && mhi - mlo + 1 >= a.min_wf_len) {   // (uniform: every thread read the same limits)
          const int dmin = min(TRmin[11], max(plen, tlen));   // (collected in the pass above)
          int fk = INT_MAX, lk = INT_MIN;
          if (my_cells <= 1) {   // the usual case: this thread's one cell is still in registers
            if (my_cells == 1 && my_off >= 0 && max(tlen, plen + my_k) - my_off - dmin <= a.max_dist_thr) fk = lk = my_k;
          } else {
            for (int k = mlo + tid; k <= mhi; k += T) {
              const int off = wR[0][k];
              if (off >= 0 && max(tlen, plen + k) - off - dmin <= a.max_dist_thr) { fk = min(fk, k); lk = max(lk, k); }
            }
          }
          fk = wide_wave_min(fk); lk = wide_wave_max(lk);
          if (lane == 0 && fk != INT_MAX) { atomicMin(&TRmin[12], fk); atomicMax(&TRmin[13], lk); }
          __syncthreads();
          const int lc = TRmin[12], hc = TRmin[13];   // (INT_MAX / INT_MIN: no diagonal qualifies)
          int new_lo = mlo, new_hi = mhi;
          const int top_limit = min(ak, mhi);
          if (top_limit > mlo) new_lo = min(lc, top_limit);
          const int bottom_limit = max(ak, new_lo);
          if (bottom_limit < mhi) new_hi = max(hc, bottom_limit);
          steps_wait = a.steps_between;
          if (new_lo != mlo || new_hi != mhi) {   // (the dropped cells read NULL from now on: the limits say so)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              tlo[c] = max(tlo[c], new_lo); thi[c] = min(thi[c], new_hi);
              if (tlo[c] > thi[c]) { tlo[c] = 1; thi[c] = -1; }
            }
          }
        }
      }
      // every wave notes the limits of the rows written in this step itself (the same values from every wave): it reads them back in the
      // next steps without waiting for another wave
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { rlo[rW[c]] = tlo[c]; rhi[rW[c]] = thi[c]; }
      }
      if (end_reason) break;
    }
    if (hand_on) end_reason = 3;

    // =============================== finish ===============================
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
          // walk the origin codes back from the end cell, then unpack forwards re-extending the matches (wfa_general.hpp PB)
          const long long n = wide_walk_unpack<TWO>(hist, a.hist_stride, pb_codes, pb_codes + pb_used, pb_cap - pb_used - (long long)(end_t + 2) * 12,
                                                    end_t, end_k, a.X, a.OE, a.E, a.OE2, a.E2, sP, sT, plen, tlen, a.cigar_ops + a.cigar_off[pair]);
          if (n < 0) {
            out_status = WFA_INTERNAL_FALLBACK; out_score = 0;
            a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;
          } else {
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

// host entry point (csrc/k_wide.hip): two = gap-affine-2p with the rows in the HBM workspace
int launch_wide(bool full, bool two, bool grows, const WideArgs& a, int grid, int threads, size_t smem, hipStream_t stream, bool w32 = false);

}  // namespace wfa
