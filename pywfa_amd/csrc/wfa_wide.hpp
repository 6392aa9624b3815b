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
// Scope: gap-affine, match = 0, 2-bit pairs, no heuristic, end-to-end or ends-free, score-only or full CIGAR.  Full CIGAR
// keeps the piggy-back history of the general kernel (one byte of origin codes per cell + a 12-byte directory record per
// step in the workgroup's slice of the HBM workspace; wfa_general.hpp PB) and the same walk / forward unpack.
// A pair whose wavefront leaves the rows, whose history does not fit, or that runs into an all-NULL stretch is handed on
// (fb_list) to the general kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
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
  int g, X, OE, E;        // score step and the penalties in steps
  int ef, pbf, pef, tbf, tef;
  int max_steps;
  int wcap;               // diagonals per row
  int seq_words;          // LDS words per sequence (>= words of the longest sequence + 3)
};

#define WFA_WIDE_NULL (-16384)
#define WFA_WIDE_CTRL_INTS 32

__device__ __forceinline__ uint32_t wide_ffbl(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }

static inline size_t wide_smem_bytes(int X, int OE, int E, int wcap, int seq_words) {
  const int NM = (X > OE ? X : OE) + 1, NG = E + 1, NR = NM + 2 * NG + 1;   // + the always-NULL row
  const size_t rw = (size_t)((wcap + 2 + 1) & ~1);
  return (size_t)(WFA_WIDE_CTRL_INTS + 2 * NR) * 4 + (size_t)2 * seq_words * 4 + (size_t)NR * rw * 2;
}

template <bool FULL>
__global__ void __launch_bounds__(1024)
wfa_wide_kernel(const WideArgs a) {
  extern __shared__ int wsm[];
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
  const int DM = max(a.X, a.OE);
  const int NM = DM + 1, NG = a.E + 1, NR = NM + 2 * NG + 1;
  const int rw = (a.wcap + 2 + 1) & ~1;            // halfs per row: guard, wcap diagonals, guard (+ pad)
  int* const ctrl = wsm;                           // [0..5] trim min x3 / max x3 (parity 0), [6..11] parity 1, [12..13] end k
  int* const rlo = wsm + WFA_WIDE_CTRL_INTS;       // trimmed limits of every row
  int* const rhi = rlo + NR;
  uint32_t* const sP = reinterpret_cast<uint32_t*>(rhi + NR);
  uint32_t* const sT = sP + a.seq_words;
  short* const rows = reinterpret_cast<short*>(sT + a.seq_words);
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
    __syncthreads();   // the previous pair is done with LDS
    // ---- sequences and rows ----
    {
      const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
      const uint32_t* gp = a.words + pm.p_woff;
      const uint32_t* gt = a.words + pm.t_woff;
      for (int i = tid; i < a.seq_words; i += T) { sP[i] = (i < nwp) ? gp[i] : 0u; sT[i] = (i < nwt) ? gt[i] : 0u; }
      uint32_t* r32 = reinterpret_cast<uint32_t*>(rows);
      const int n32 = NR * rw / 2;
      for (int i = tid; i < n32; i += T) r32[i] = 0xC000C000u;   // NULL, NULL
      for (int i = tid; i < NR; i += T) { rlo[i] = 1; rhi[i] = -1; }
      if (tid < 14) ctrl[tid] = (tid >= 12 || (tid % 6) < 3) ? INT_MAX : INT_MIN;
    }
    bool hand_on = (plen + tlen > 32000) || (-pbf < kmin) || (tbf > kmax) || (ak < kmin) || (ak > kmax);
    int end_reason = 0;   // 1 reached, 3 handed on, 4 step limit
    int end_k = 0, end_off = 0, end_t = 0;
    long long pb_used = 0;                                  // FULL: code bytes in use
    uint8_t* const pb_codes = FULL ? reinterpret_cast<uint8_t*>(hist) : nullptr;
    const long long pb_cap = FULL ? a.hist_stride * 4 : 0;  // bytes shared by codes (bottom-up) and directory (top-down)
    int null_run = 0;
    __syncthreads();

    for (int t = 0; !hand_on; ++t) {
      const int s = t * a.g;
      const int par = t & 1;
      int* const TRmin = ctrl + 6 * par;
      int* const TRmax = TRmin + 3;
      // the limit is tested after compute-next of a score and before its extension (R/wavefront_unialign.c:98-107)
      if (t > 0 && s >= a.max_steps) { end_reason = 4; break; }
      if (t > 16000) { end_reason = 3; break; }
      // ---- rows of this step and its inputs ----
      const int rM = t % NM, rI = NM + t % NG, rD = NM + NG + t % NG;
      const int iX = (t >= a.X) ? (t - a.X) % NM : NULLROW;
      const int iO = (t >= a.OE) ? (t - a.OE) % NM : NULLROW;
      const int iI = (t >= a.E) ? NM + (t - a.E) % NG : NULLROW;
      const int iD = (t >= a.E) ? NM + NG + (t - a.E) % NG : NULLROW;
      int lo, hi;
      if (t == 0) { lo = -pbf; hi = tbf; }
      else {
        // R/wavefront_compute.c:40-86 (a null input counts with lo = 1, hi = -1, as there)
        lo = min(min(rlo[iX], rlo[iO] - 1), min(rlo[iI] + 1, rlo[iD] - 1));
        hi = max(max(rhi[iX], rhi[iO] + 1), max(rhi[iI] + 1, rhi[iD] - 1));
        const bool all_null = rlo[iX] > rhi[iX] && rlo[iO] > rhi[iO] && rlo[iI] > rhi[iI] && rlo[iD] > rhi[iD];
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
      short* const wM = rows + rM * rw + koff;
      short* const wI = rows + rI * rw + koff;
      short* const wD = rows + rD * rw + koff;
      const short* const pX = rows + iX * rw + koff;
      const short* const pO = rows + iO * rw + koff;
      const short* const pI = rows + iI * rw + koff;
      const short* const pD = rows + iD * rw + koff;
      // stale cells of the rows written now (their previous wavefronts) outside the range written below
      {
        const int olo[3] = {rlo[rM], rlo[rI], rlo[rD]}, ohi[3] = {rhi[rM], rhi[rI], rhi[rD]};
        short* const w3[3] = {wM, wI, wD};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (olo[c] > ohi[c]) continue;
          if (lo > hi) { for (int k = olo[c] + tid; k <= ohi[c]; k += T) w3[c][k] = (short)WFA_WIDE_NULL; continue; }
          for (int k = olo[c] + tid; k <= min(ohi[c], lo - 1); k += T) w3[c][k] = (short)WFA_WIDE_NULL;
          for (int k = max(olo[c], hi + 1) + tid; k <= ohi[c]; k += T) w3[c][k] = (short)WFA_WIDE_NULL;
        }
      }
      // ---- the pass: compute, clamp, extend, store; trimmed limits by wave ballots ----
      int wmin[3] = {INT_MAX, INT_MAX, INT_MAX}, wmax[3] = {INT_MIN, INT_MIN, INT_MIN};   // (wave-uniform)
      for (int k0 = lo + (tid & ~63); k0 <= hi; k0 += T) {
        const int k = k0 + lane;
        const bool in = k <= hi;
        int mv = WFA_WIDE_NULL, iv = WFA_WIDE_NULL, dv = WFA_WIDE_NULL, code = 0;
        if (in) {
          if (t == 0) {
            mv = max(k, 0);   // R/wavefront_aligner.c:251-310: offset 0 on diagonal 0, the free begins on theirs
          } else {
            const int mo_lo = pO[k - 1], ie_lo = pI[k - 1], mo_hi = pO[k + 1], de_hi = pD[k + 1];
            iv = max(mo_lo, ie_lo) + 1;
            dv = max(mo_hi, de_hi);
            const int x1 = pX[k] + 1;
            mv = max(dv, max(x1, iv));
            if (FULL) {
              // the backtrace's choice on equal offsets (R/wavefront_backtrace.c:49-59), as wfa_general.hpp PB
              const int mc = (x1 >= max(dv, iv)) ? 0 : ((dv >= iv) ? 1 : 2);
              code = mc | ((ie_lo >= mo_lo) ? 4 : 0) | ((de_hi >= mo_hi) ? 8 : 0);
            }
          }
        }
        // in bounds: max(k, 0) <= offset <= min(tlen, plen + k) (0 <= h <= tlen and 0 <= v <= plen)
        const int limk = min(tlen, plen + k);   // the largest offset on diagonal k
        const int base = max(k, 0);
        const bool kin = in && limk >= base;    // (a diagonal beyond -plen .. tlen holds no cell)
        const uint32_t span = (uint32_t)(limk - base);
        const bool m_in = (uint32_t)(mv - base) <= span && kin;
        const bool i_in = (uint32_t)(iv - base) <= span && kin;
        const bool d_in = (uint32_t)(dv - base) <= span && kin;
        if (!m_in) mv = WFA_WIDE_NULL;          // only M is clamped (R/wavefront_compute_affine.c:80-84)
        const unsigned long long bm = __ballot(m_in), bi = __ballot(i_in), bd = __ballot(d_in);
        if (bm) { wmin[0] = min(wmin[0], k0 + (int)__builtin_ctzll(bm)); wmax[0] = max(wmax[0], k0 + 63 - (int)__builtin_clzll(bm)); }
        if (bi) { wmin[1] = min(wmin[1], k0 + (int)__builtin_ctzll(bi)); wmax[1] = max(wmax[1], k0 + 63 - (int)__builtin_clzll(bi)); }
        if (bd) { wmin[2] = min(wmin[2], k0 + (int)__builtin_ctzll(bd)); wmax[2] = max(wmax[2], k0 + 63 - (int)__builtin_clzll(bd)); }
        // extend M (R/wavefront_extend_kernels.c:64-88): 32 bases per round, never past either sequence end
        if (bm) {
          int h = mv, v = mv - k, left = m_in ? limk - mv : 0;
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
              const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)h << 1);
              const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
              const uint32_t fb = min(wide_ffbl(xl), wide_ffbl(xh) | 32u);   // (v_ffbl_b32 gives ~0 for 0)
              const int m = min((int)(fb >> 1), min(32, left));
              v += m; h += m; left -= m;
              more = (m == 32) && (left > 0);
            }
          }
          if (m_in) {
            mv = h;
            // termination on the extended offset
            if (a.ef) {
              if ((h >= tlen && plen - v <= a.pef) || (v >= plen && tlen - h <= a.tef)) atomicMin(&ctrl[12 + par], k);
            } else if (k == ak && h >= tlen) {
              ctrl[12 + par] = k;
            }
          }
        }
        if (in) {
          // (negative I / D values are stored as they are: they start at NULL and gain at most 1 per step, so they stay
          // negative for the 16 000 steps a pair may take here, and a negative offset is never in bounds)
          wM[k] = (short)mv;
          wI[k] = (short)iv;
          wD[k] = (short)dv;
          if (FULL) pb_codes[code_base + (k - lo)] = (uint8_t)code;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (wmin[c] != INT_MAX) { atomicMin(&TRmin[c], wmin[c]); atomicMax(&TRmax[c], wmax[c]); }
      }
      __syncthreads();   // rows, trimmed limits and the end flag of this step are visible
      // ---- trimmed limits (R/wavefront_compute.c:571-605): first / last in-bounds cell; none -> null ----
      int tlo[3], thi[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int mn = TRmin[c], mx = TRmax[c];
        if (mn != INT_MAX) { tlo[c] = mn; thi[c] = mx; } else { tlo[c] = 1; thi[c] = -1; }
      }
      const int ek = ctrl[12 + par];
      if (ek != INT_MAX) { end_reason = 1; end_k = ek; end_off = wM[ek]; end_t = t; }
      // I / D cells outside their trimmed limits become NULL (M's are NULL already)
      if (lo <= hi) {
        short* const w2[2] = {wI, wD};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int l = tlo[c + 1], h2 = thi[c + 1];
          if (l > h2) { for (int k = lo + tid; k <= hi; k += T) w2[c][k] = (short)WFA_WIDE_NULL; }
          else {
            for (int k = lo + tid; k < l; k += T) w2[c][k] = (short)WFA_WIDE_NULL;
            for (int k = h2 + 1 + tid; k <= hi; k += T) w2[c][k] = (short)WFA_WIDE_NULL;
          }
        }
      }
      if (tid == 0) {
        rlo[rM] = tlo[0]; rhi[rM] = thi[0]; rlo[rI] = tlo[1]; rhi[rI] = thi[1]; rlo[rD] = tlo[2]; rhi[rD] = thi[2];
        // the other parity's scratch for the next step
        int* o = ctrl + 6 * (par ^ 1);
        o[0] = INT_MAX; o[1] = INT_MAX; o[2] = INT_MAX; o[3] = INT_MIN; o[4] = INT_MIN; o[5] = INT_MIN;
        ctrl[12 + (par ^ 1)] = INT_MAX;
      }
      __syncthreads();
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
            const int src = (comp == 0) ? ((cd & 3) == 0 ? 0 : ((cd & 3) == 1 ? 1 : 3)) : (comp == 1 ? 3 : 1);   // 0 X, 1 D, 3 I
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

// host entry point (csrc/k_wide.hip)
int launch_wide(bool full, const WideArgs& a, int grid, int threads, size_t smem, hipStream_t stream);

}  // namespace wfa
