// wfa_general.hpp — the general wavefront-alignment kernel: ONE alignment per workgroup at a time,
// lanes over diagonals k, any sequence length, gap-affine and gap-affine-2p, end-to-end and
// ends-free, none / wf-adaptive / X-drop, score-only (modular ring of max_score_scope wavefronts)
// or full CIGAR (explicit wavefront history + on-device backtrace).
//
// What each phase restates (R = /root/reference/pywfa/WFA2_lib/wavefront):
//   driver loop            R/wavefront_unialign.c:241-273 (extend -> finished? -> ++score -> compute -> limits)
//   wavefront 0            R/wavefront_aligner.c:251-310
//   extend                 R/wavefront_extend_kernels.c:64-163, R/wavefront_extend.c:90-125,263-297
//   termination            R/wavefront_termination.c:37-61 (end2end), :115-162 (ends-free, lowest k wins)
//   heuristic cut-off      R/wavefront_heuristic.c:176-293 (wf-adaptive), :297-383 (X-drop), :509-567
//   compute-next           R/wavefront_compute_affine.c:44-86,229-260, R/wavefront_compute_affine2p.c:45-106,334-368
//                          with R/wavefront_compute.c:40-86 (limits), :298-344 (inputs by score), :571-605 (trim)
//   finish / status        R/wavefront_unialign.c:98-107,147-237
//   backtrace              R/wavefront_backtrace.c:49-101,320-529
//
// Storage.  A workgroup owns one slice of the HBM workspace (a.ws + blockIdx.x * a.ws_stride int32):
//   score-only: a ring of `scope` slots x NCOMP components x (plen+tlen+3) offsets, indexed by k;
//   full:       an arena; the offsets of score s (NCOMP x width) are appended bottom-up, the
//               per-score directory record {lo[],hi[],base,width,data,exists} grows top-down.
// The directory records of the last `scope` scores are mirrored in LDS (the "meta ring"); reading a
// wavefront outside its [lo,hi] yields NULL exactly like the reference's lazily padded arrays
// (R/wavefront_compute.c:490-567).  Integer arithmetic only.
#pragma once
#include "wfa_rtc_compat.hpp"
#include "wfa_common.hpp"
#include "wfa_hip.h"

namespace wfa {

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}

// 16 bases starting at base position `pos` of a 2-bit packed sequence (bits beyond the sequence end
// are garbage; callers clamp by the remaining length).
__device__ __forceinline__ uint32_t window16(const uint32_t* __restrict__ w, int pos) {
  const int i = pos >> 4;
  const uint32_t lo = w[i];
  const uint32_t hi = w[i + 1];
  return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(pos & 15) << 1);
}

// Extend one diagonal (R/wavefront_extend_kernels.c:64-88): length of the common prefix of
// pattern[v..] and text[h..], which cannot pass either sequence end (the reference's sentinels).
template <bool PACKED>
struct SeqView {
  const uint32_t* pw;
  const uint32_t* tw;
  const uint8_t* pb;
  const uint8_t* tb;
  int plen, tlen, wildcard;

  __device__ __forceinline__ int extend(int k, int off) const {
    const int h = off, v = off - k;
    const int maxrun = min(plen - v, tlen - h);
    int run = 0;
    if (PACKED) {
      while (run < maxrun) {
        const uint32_t x = window16(pw, v + run) ^ window16(tw, h + run);
        const int m = x ? (__builtin_ctz(x) >> 1) : 16;
        run += m;
        if (m < 16) break;
      }
      run = min(run, maxrun);
    } else {
      if (wildcard < 0) {
        while (run < maxrun && pb[v + run] == tb[h + run]) ++run;
      } else {
        while (run < maxrun) {
          const int pc = pb[v + run], tc = tb[h + run];
          if (!(pc == tc || pc == wildcard || tc == wildcard)) break;
          ++run;
        }
      }
    }
    return off + run;
  }
};

// one input wavefront as the compute loop sees it: value(k) = (lo<=k<=hi) ? ws[idx0 + k] : NULL
struct WfIn {
  int lo, hi, idx0;
  __device__ __forceinline__ bool null() const { return lo > hi; }
  __device__ __forceinline__ int get(const int* __restrict__ ws, int k) const {
    return (k >= lo && k <= hi) ? ws[idx0 + k] : WFA_OFFSET_NULL;
  }
};

template <int NCOMP>
struct Meta {
  // layout of one directory record / meta-ring entry (ints)
  static constexpr int LO = 0;          // lo[NCOMP]
  static constexpr int HI = NCOMP;      // hi[NCOMP]
  static constexpr int BASE = 2 * NCOMP;
  static constexpr int WIDTH = 2 * NCOMP + 1;
  static constexpr int DATA = 2 * NCOMP + 2;
  static constexpr int EXISTS = 2 * NCOMP + 3;  // M "pointer != NULL" of the reference
  static constexpr int INTS = 2 * NCOMP + 4;
};

template <int NCOMP>
__device__ __forceinline__ WfIn fetch_in(const int* ring, int scope, int s, int c) {
  typedef Meta<NCOMP> MT;
  WfIn in;
  in.lo = 1; in.hi = -1; in.idx0 = 0;
  if (s >= 0) {
    const int* m = ring + (s % scope) * MT::INTS;
    const int lo = m[MT::LO + c], hi = m[MT::HI + c];
    if (lo <= hi) {
      in.lo = lo; in.hi = hi;
      in.idx0 = m[MT::DATA] + c * m[MT::WIDTH] - m[MT::BASE];
    }
  }
  in.lo = __builtin_amdgcn_readfirstlane(in.lo);
  in.hi = __builtin_amdgcn_readfirstlane(in.hi);
  in.idx0 = __builtin_amdgcn_readfirstlane(in.idx0);
  return in;
}

// candidate of the backtrace (R/wavefront_backtrace.c:64-219), read from the HBM directory
template <int NCOMP>
__device__ __forceinline__ long long bt_cand(const int* ws, long long ws_stride, int s, int c, int k,
                                             int add, int type) {
  typedef Meta<NCOMP> MT;
  if (s < 0) return WFA_OFFSET_NULL;
  const int* m = ws + ws_stride - (long long)(s + 1) * MT::INTS;
  if (k < m[MT::LO + c] || k > m[MT::HI + c]) return WFA_OFFSET_NULL;
  const int o = ws[m[MT::DATA] + c * m[MT::WIDTH] + (k - m[MT::BASE])];
  return (((long long)(o + add)) << 4) | type;
}

struct OpsWriter {
  uint8_t* buf;       // region of this pair
  long long begin;    // index (relative to buf) of the first valid op; ops are written right-to-left
  __device__ __forceinline__ void push(char c, int n) {
    while (n-- > 0) buf[--begin] = (uint8_t)c;
  }
};

// R/wavefront_backtrace.c:223-319 (single-component metrics), single lane.
__device__ inline void backtrace_linear(const int* ws, long long ws_stride, const WfaDevConfig& cfg, int plen, int tlen,
                                        int end_s, int end_k, int end_off, OpsWriter& ops) {
  enum { BT_I1_OPEN = 1, BT_D1_OPEN = 5, BT_M = 9 };
  int s = end_s, k = end_k, offset = end_off;
  int h = offset, v = offset - k;
  if (v < plen) ops.push('D', plen - v);
  if (h < tlen) ops.push('I', tlen - h);
  while (v > 0 && h > 0 && s > 0) {
    const int s_x = s - cfg.x, s_o = s - cfg.o1;
    long long best = (cfg.metric != 0) ? bt_cand<1>(ws, ws_stride, s_x, 0, k, 1, BT_M) : (long long)WFA_OFFSET_NULL;
    best = max(best, bt_cand<1>(ws, ws_stride, s_o, 0, k - 1, 1, BT_I1_OPEN));
    best = max(best, bt_cand<1>(ws, ws_stride, s_o, 0, k + 1, 0, BT_D1_OPEN));
    if (best < 0) break;
    const int src = (int)(best >> 4);
    ops.push('M', offset - src);
    offset = src;
    v = offset - k; h = offset;
    if (v <= 0 || h <= 0) break;
    const int type = (int)(best & 0xF);
    if (type == BT_M) { s = s_x; ops.push('X', 1); --offset; }
    else if (type == BT_I1_OPEN) { s = s_o; ops.push('I', 1); --k; --offset; }
    else { s = s_o; ops.push('D', 1); ++k; }
    v = offset - k; h = offset;
  }
  if (v > 0 && h > 0) { const int n = min(v, h); ops.push('M', n); v -= n; h -= n; }
  ops.push('D', max(v, 0));
  ops.push('I', max(h, 0));
}

// R/wavefront_backtrace.c:320-529, single lane.
template <int NCOMP>
__device__ void backtrace(const int* ws, long long ws_stride, const WfaDevConfig& cfg, int plen, int tlen,
                          int end_s, int end_k, int end_off, OpsWriter& ops) {
  enum { BT_I1_OPEN = 1, BT_I1_EXT, BT_I2_OPEN, BT_I2_EXT, BT_D1_OPEN, BT_D1_EXT, BT_D2_OPEN, BT_D2_EXT, BT_M };
  int comp = 0, s = end_s, k = end_k, offset = end_off;
  int h = offset, v = offset - k;
  if (v < plen) ops.push('D', plen - v);
  if (h < tlen) ops.push('I', tlen - h);
  while (v > 0 && h > 0 && s > 0) {
    const int s_x = s - cfg.x, s_o1 = s - cfg.o1 - cfg.e1, s_e1 = s - cfg.e1;
    const int s_o2 = s - cfg.o2 - cfg.e2, s_e2 = s - cfg.e2;
    long long best;
    if (comp == 0) {
      best = bt_cand<NCOMP>(ws, ws_stride, s_x, 0, k, 1, BT_M);
      best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_o1, 0, k - 1, 1, BT_I1_OPEN));
      best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_e1, 1, k - 1, 1, BT_I1_EXT));
      best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_o1, 0, k + 1, 0, BT_D1_OPEN));
      best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_e1, 2, k + 1, 0, BT_D1_EXT));
      if (NCOMP == 5) {
        best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_o2, 0, k - 1, 1, BT_I2_OPEN));
        best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_e2, 3, k - 1, 1, BT_I2_EXT));
        best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_o2, 0, k + 1, 0, BT_D2_OPEN));
        best = max(best, bt_cand<NCOMP>(ws, ws_stride, s_e2, 4, k + 1, 0, BT_D2_EXT));
      }
    } else if (comp == 1) {
      best = max(bt_cand<NCOMP>(ws, ws_stride, s_o1, 0, k - 1, 1, BT_I1_OPEN),
                 bt_cand<NCOMP>(ws, ws_stride, s_e1, 1, k - 1, 1, BT_I1_EXT));
    } else if (comp == 2) {
      best = max(bt_cand<NCOMP>(ws, ws_stride, s_o1, 0, k + 1, 0, BT_D1_OPEN),
                 bt_cand<NCOMP>(ws, ws_stride, s_e1, 2, k + 1, 0, BT_D1_EXT));
    } else if (comp == 3) {
      best = max(bt_cand<NCOMP>(ws, ws_stride, s_o2, 0, k - 1, 1, BT_I2_OPEN),
                 bt_cand<NCOMP>(ws, ws_stride, s_e2, NCOMP == 5 ? 3 : 1, k - 1, 1, BT_I2_EXT));
    } else {
      best = max(bt_cand<NCOMP>(ws, ws_stride, s_o2, 0, k + 1, 0, BT_D2_OPEN),
                 bt_cand<NCOMP>(ws, ws_stride, s_e2, NCOMP == 5 ? 4 : 2, k + 1, 0, BT_D2_EXT));
    }
    if (best < 0) break;
    if (comp == 0) {
      const int src = (int)(best >> 4);
      ops.push('M', offset - src);
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
    if (type == BT_M) { ops.push('X', 1); --offset; }
    else if (type <= BT_I2_EXT) { ops.push('I', 1); --k; --offset; }
    else { ops.push('D', 1); ++k; }
    v = offset - k; h = offset;
  }
  if (comp == 0) {
    if (v > 0 && h > 0) {
      const int n = min(v, h);
      ops.push('M', n);
      v -= n; h -= n;
    }
    ops.push('D', max(v, 0));
    ops.push('I', max(h, 0));
  }
}

// R/wavefront_compute.c:108-120 with WF_SCORE_TO_SW_SCORE (R/wavefront_penalties.h:73)
__device__ __forceinline__ int classic_score(const WfaDevConfig& cfg, int v, int h, int s) {
  if (cfg.metric <= 1) return s;  // indel / edit distances are reported as they are
  if (cfg.match == 0) return -s;
  return ((-cfg.match) * (v + h) - s) / 2;
}

// PB (with FULL): the piggy-back form of the history (R/wavefront_backtrace_offload.c, R/wavefront_pcigar.c): the
// offsets live in the score-only ring, the history keeps ONE BYTE of origin codes per (score, diagonal) — which candidate made M
// (gap-affine: mismatch / deletion / insertion; 2p: mismatch / D1 / D2 / I1 / I2, in the backtrace's priority order) and
// open-or-extend for every gap component — plus a 12-byte directory record per score; the walk follows the codes back from the
// end cell and the op string is unpacked forwards, re-extending the matches (a wavefront cell is always extended to its end).
// 20x less history than the explicit arena for gap-affine-2p (exact C4: ~0.4 GB -> ~20 MB per pair).
template <int NCOMP, bool PACKED, bool FULL, bool PB = false>
__global__ void __launch_bounds__(512)
wfa_general_kernel(const WfaKernelArgs a) {
  static_assert(!PB || (FULL && NCOMP >= 3), "piggy-back history: full scope, gap-affine / gap-affine-2p");
  constexpr bool RING = !FULL || PB;   // offsets in the modular ring of `scope` wavefronts
  typedef Meta<NCOMP> MT;
  extern __shared__ int smem[];
  const WfaDevConfig& cfg = a.cfg;
  const int scope = cfg.scope;
  int* const ring = smem;                   // scope * MT::INTS
  int* const TR = smem + scope * MT::INTS;  // trim scratch: min k in-bounds [NCOMP], max k in-bounds [NCOMP]
  int* const EK = TR + 2 * NCOMP;           // EK[0] end k; EK[1] heur min; EK[2] lo cand; EK[3] hi cand; EK[4] max
  const int tid = threadIdx.x, T = blockDim.x;
  int* const ws = a.ws + (long long)blockIdx.x * a.ws_stride;
  const long long ws_stride = a.ws_stride;
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  const uint32_t wb = a.wbeg_dev ? *a.wbeg_dev : 0u;

  for (uint32_t wi = wb + blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    const WfaPairMeta pm = a.meta[pair];
    const int plen = pm.plen, tlen = pm.tlen;
    SeqView<PACKED> seq;
    seq.plen = plen; seq.tlen = tlen; seq.wildcard = cfg.wildcard;
    if (PACKED) {
      seq.pw = a.words + pm.p_woff; seq.tw = a.words + pm.t_woff; seq.pb = nullptr; seq.tb = nullptr;
    } else {
      seq.pb = a.bytes + a.p_boff[pair]; seq.tb = a.bytes + a.t_boff[pair]; seq.pw = nullptr; seq.tw = nullptr;
    }
    const int pbf = (cfg.endsfree && cfg.match == 0) ? cfg.pbf : 0;
    const int tbf = (cfg.endsfree && cfg.match == 0) ? cfg.tbf : 0;
    const int rstride = plen + tlen + 3;   // ring: offsets of k in [-plen-1, tlen+1]
    const int rbase = -plen - 1;
    bool overflow = false;
    const long long ring_ints = (long long)scope * NCOMP * rstride;
    if (RING && ring_ints + (PB ? 64 : 0) > ws_stride) overflow = true;
    if (FULL && !PB && (long long)(tbf + pbf + 1) * NCOMP + 2 * MT::INTS > ws_stride) overflow = true;
    // PB: code bytes grow upwards from the end of the ring, directory records {lo, hi, byte base} downwards from the top
    uint8_t* const pb_codes = PB ? reinterpret_cast<uint8_t*>(ws + ring_ints) : nullptr;
    const long long pb_cap = PB ? (ws_stride - ring_ints) * 4 : 0;   // bytes shared by codes and directory
    long long pb_used = 0;                                            // code bytes in use

    // ---- wavefront 0 ----
    int used = 0;  // FULL: ints of the arena in use
    int cur_lo = -pbf, cur_hi = tbf, cur_idx0, cur_exists = 1;
    {
      const int base = !RING ? -pbf : rbase;
      const int width = !RING ? (tbf + pbf + 1) : rstride;
      const int data = 0;
      cur_idx0 = data - base;
      __syncthreads();  // previous pair fully done with ring / scratch
      if (tid == 0) {
        int* m = ring;
        for (int c = 0; c < NCOMP; ++c) { m[MT::LO + c] = 1; m[MT::HI + c] = -1; }
        m[MT::LO] = -pbf; m[MT::HI] = tbf;
        m[MT::BASE] = base; m[MT::WIDTH] = width; m[MT::DATA] = data; m[MT::EXISTS] = 1;
        for (int c = 0; c < 2 * NCOMP; ++c) TR[c] = (c < NCOMP) ? INT_MAX : INT_MIN;
        EK[0] = INT_MAX; EK[1] = INT_MAX; EK[2] = INT_MAX; EK[3] = INT_MIN; EK[4] = INT_MIN;
      }
      if (!overflow) {
        for (int k = -pbf + tid; k <= tbf; k += T) ws[cur_idx0 + k] = (k > 0) ? k : 0;
      }
      if (FULL && !PB) used = NCOMP * width;
      if (PB && !overflow && tid == 0) {   // score 0 has no origin: a record with an empty range
        int* d = ws + ws_stride - 3;
        d[0] = 1; d[1] = 0; d[2] = 0;
      }
    }

    int s = 0, null_steps = 0;
    // match < 0 with free begins: the begin-free cells enter the M wavefront of score j * (-match) (R/wavefront_compute.c:124-254);
    // score scope only (with a backtrace the reference fails on ordinary inputs: refused by wfa_hip_config_validate)
    const bool ef_seed = !FULL && cfg.match != 0 && cfg.endsfree && (cfg.pbf != 0 || cfg.tbf != 0);
    int steps_wait = cfg.steps_between, have_max_sw = 0, max_sw = 0;
    int end_reason = overflow ? 3 : 0;  // 1 reached, 2 unreachable, 3 overflow, 4 max steps
    int end_k = 0, end_off = WFA_OFFSET_NULL;

    while (!end_reason) {
      // =============================== extend(s) ===============================
      __syncthreads();  // offsets of score s (and the meta ring) are visible; TR reads of compute(s) are done
      if (tid == 0) {
        for (int c = 0; c < 2 * NCOMP; ++c) TR[c] = (c < NCOMP) ? INT_MAX : INT_MIN;
      }
      if (!cur_exists) {
        if (null_steps > scope) { end_reason = 2; break; }
      } else {
        for (int k = cur_lo + tid; k <= cur_hi; k += T) {
          const int off = ws[cur_idx0 + k];
          if (off == WFA_OFFSET_NULL) continue;
          const int ext = seq.extend(k, off);
          if (ext != off) ws[cur_idx0 + k] = ext;
          if (cfg.endsfree) {
            const int h = ext, v = ext - k;
            if ((h >= tlen && plen - v <= cfg.pef) || (v >= plen && tlen - h <= cfg.tef)) atomicMin(&EK[0], k);
          }
        }
        __syncthreads();
        if (cfg.endsfree) {
          const int ek = EK[0];
          if (ek != INT_MAX) { end_reason = 1; end_k = ek; end_off = ws[cur_idx0 + ek]; }
        } else {
          const int ak = tlen - plen;
          if (cur_lo <= ak && ak <= cur_hi && ws[cur_idx0 + ak] >= tlen) { end_reason = 1; end_k = ak; end_off = tlen; }
        }
        if (end_reason) break;
        // ---------------------------- heuristic cut-off ----------------------------
        if (cfg.heuristic != 0 && cur_lo <= cur_hi) {
          --steps_wait;
          int new_lo = cur_lo, new_hi = cur_hi;
          if (cfg.heuristic == 1) {
            if (steps_wait <= 0 && (cur_hi - cur_lo + 1) >= cfg.min_wf_len) {
              int dmin = max(plen, tlen);
              for (int k = cur_lo + tid; k <= cur_hi; k += T) {
                const int off = ws[cur_idx0 + k];
                const int d = (off >= 0) ? max(plen - (off - k), tlen - off) : -WFA_OFFSET_NULL;
                dmin = min(dmin, d);
              }
              dmin = wave_min(dmin);
              if ((tid & 63) == 0) atomicMin(&EK[1], dmin);
              __syncthreads();
              dmin = EK[1];
              int lc = INT_MAX, hc = INT_MIN;
              for (int k = cur_lo + tid; k <= cur_hi; k += T) {
                const int off = ws[cur_idx0 + k];
                const int d = (off >= 0) ? max(plen - (off - k), tlen - off) : -WFA_OFFSET_NULL;
                if (d - dmin <= cfg.max_dist_thr) { lc = min(lc, k); hc = max(hc, k); }
              }
              lc = wave_min(lc); hc = wave_max(hc);
              if ((tid & 63) == 0) { atomicMin(&EK[2], lc); atomicMax(&EK[3], hc); }
              __syncthreads();
              lc = EK[2]; hc = EK[3];
              const int ak = tlen - plen;
              const int top_limit = min(ak, cur_hi);
              if (top_limit > cur_lo) new_lo = min(lc, top_limit);
              const int bottom_limit = max(ak, new_lo);
              if (bottom_limit < cur_hi) new_hi = max(hc, bottom_limit);
              steps_wait = cfg.steps_between;
            }
          } else if (cfg.heuristic == 2) {
            if (steps_wait <= 0) {
              const int g = (cfg.match != 0) ? -cfg.match : -1;  // R/wavefront_heuristic.c:306-307
              int cmax = INT_MIN, lc = INT_MAX, hc = INT_MIN;
              for (int k = cur_lo + tid; k <= cur_hi; k += T) {
                const int off = ws[cur_idx0 + k];
                if (off < 0) continue;
                const int sw = (g * ((off - k) + off) - s) / 2;
                cmax = max(cmax, sw);
                if (have_max_sw && max_sw - sw < cfg.xdrop) { lc = min(lc, k); hc = max(hc, k); }
              }
              cmax = wave_max(cmax); lc = wave_min(lc); hc = wave_max(hc);
              if ((tid & 63) == 0) { atomicMax(&EK[4], cmax); atomicMin(&EK[2], lc); atomicMax(&EK[3], hc); }
              __syncthreads();
              cmax = EK[4]; lc = EK[2]; hc = EK[3];
              if (have_max_sw) {
                if (lc == INT_MAX) { new_lo = cur_hi + 1; new_hi = cur_hi; }
                else { new_lo = lc; new_hi = hc; }
                if (cmax > max_sw) max_sw = cmax;
              } else {
                max_sw = cmax; have_max_sw = 1;
              }
              steps_wait = cfg.steps_between;
            }
          }
          if (new_lo != cur_lo || new_hi != cur_hi) {
            cur_lo = new_lo; cur_hi = new_hi;
            if (tid == 0) {
              int* m = ring + (s % scope) * MT::INTS;
              m[MT::LO] = new_lo; m[MT::HI] = new_hi;
              for (int c = 1; c < NCOMP; ++c) {  // wf_heuristic_equate (R/wavefront_heuristic.c:161-172)
                if (m[MT::LO + c] <= m[MT::HI + c]) {
                  m[MT::LO + c] = max(m[MT::LO + c], new_lo);
                  m[MT::HI + c] = min(m[MT::HI + c], new_hi);
                }
              }
            }
          }
        }
      }
      if (FULL && !PB && tid == 0) {
        // final directory record of score s (its lo/hi can no longer change)
        const int* m = ring + (s % scope) * MT::INTS;
        int* d = ws + ws_stride - (long long)(s + 1) * MT::INTS;
        for (int c = 0; c < MT::INTS; ++c) d[c] = m[c];
      }
      // =============================== compute(s+1) ===============================
      ++s;
      __syncthreads();  // meta ring updates of the cut-off are visible; EK reads are done
      if (tid == 0) { EK[0] = INT_MAX; EK[1] = INT_MAX; EK[2] = INT_MAX; EK[3] = INT_MIN; EK[4] = INT_MIN; }
      // single-component metrics (NCOMP == 1: indel / edit / gap-linear, R/wavefront_compute_edit.c:44-100,
      // R/wavefront_compute_linear.c:44-74) are the same recurrence with no I/D inputs and the gap
      // penalty in place of o+e (cfg.e1 == 0); indel has no mismatch input
      WfIn nullin; nullin.lo = 1; nullin.hi = -1; nullin.idx0 = 0;
      const WfIn mx = (NCOMP == 1 && cfg.metric == 0) ? nullin : fetch_in<NCOMP>(ring, scope, s - cfg.x, 0);
      const WfIn mo1 = fetch_in<NCOMP>(ring, scope, s - cfg.o1 - cfg.e1, 0);
      const WfIn i1e = (NCOMP == 1) ? nullin : fetch_in<NCOMP>(ring, scope, s - cfg.e1, 1);
      const WfIn d1e = (NCOMP == 1) ? nullin : fetch_in<NCOMP>(ring, scope, s - cfg.e1, 2);
      WfIn mo2 = nullin, i2e = nullin, d2e = nullin;
      if (NCOMP == 5) {
        mo2 = fetch_in<NCOMP>(ring, scope, s - cfg.o2 - cfg.e2, 0);
        i2e = fetch_in<NCOMP>(ring, scope, s - cfg.e2, 3);
        d2e = fetch_in<NCOMP>(ring, scope, s - cfg.e2, 4);
      }
      const bool all_null = mx.null() && mo1.null() && i1e.null() && d1e.null() &&
                            (NCOMP != 5 || (mo2.null() && i2e.null() && d2e.null()));
      int* const mslot = ring + (s % scope) * MT::INTS;
      if (all_null) {
        ++null_steps;
        cur_exists = 0; cur_lo = 1; cur_hi = -1; cur_idx0 = 0;
        if (FULL && !PB && (long long)used + (long long)(s + 2) * MT::INTS > ws_stride) { end_reason = 3; break; }
        if (PB) {
          if (pb_used + (long long)(s + 2) * 12 > pb_cap) { end_reason = 3; break; }
          if (tid == 0) { int* d = ws + ws_stride - 3ll * (s + 1); d[0] = 1; d[1] = 0; d[2] = 0; }
        }
        // a null step at a score that re-seeds the free begins (wavefront_compute_endsfree_allocate_null): M exists and holds
        // the cell(s) (k = j, offset j) / (k = -j, offset 0), NULL in between.  (j beyond both free begins: the reference leaves
        // lo = hi = 0 with offsets[0] unset, R/wavefront_compute.c:229-251; here the wavefront stays null)
        int seed_lo = 1, seed_hi = -1;
        if (!FULL && ef_seed && s % (-cfg.match) == 0) {
          const int ek = s / (-cfg.match);
          const bool tb = cfg.tbf >= ek, pb = cfg.pbf >= ek;
          if (tb && pb) { seed_lo = -ek; seed_hi = ek; } else if (tb) { seed_lo = seed_hi = ek; } else if (pb) { seed_lo = seed_hi = -ek; }
          if (seed_lo <= seed_hi) {
            const int data = (s % scope) * NCOMP * rstride;
            cur_exists = 1; cur_lo = seed_lo; cur_hi = seed_hi; cur_idx0 = data - rbase;
            for (int k = seed_lo + tid; k <= seed_hi; k += T) ws[cur_idx0 + k] = (k == ek && tb) ? ek : ((k == -ek && pb) ? 0 : WFA_OFFSET_NULL);
          }
        }
        if (tid == 0) {
          for (int c = 0; c < NCOMP; ++c) { mslot[MT::LO + c] = 1; mslot[MT::HI + c] = -1; }
          mslot[MT::BASE] = 0; mslot[MT::WIDTH] = 0; mslot[MT::DATA] = 0; mslot[MT::EXISTS] = 0;
          if (seed_lo <= seed_hi) {
            mslot[MT::LO] = seed_lo; mslot[MT::HI] = seed_hi;
            mslot[MT::BASE] = rbase; mslot[MT::WIDTH] = rstride; mslot[MT::DATA] = (s % scope) * NCOMP * rstride; mslot[MT::EXISTS] = 1;
          }
        }
      } else {
        null_steps = 0;
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
        int base, width, data;
        long long code_base = 0;
        if (!RING) {
          base = lo; width = hi - lo + 1; data = used;
          if ((long long)used + (long long)NCOMP * width + (long long)(s + 2) * MT::INTS > ws_stride) { end_reason = 3; break; }
          used += NCOMP * width;
        } else {
          base = rbase; width = rstride; data = (s % scope) * NCOMP * rstride;
          if (PB) {
            const long long nb = (long long)hi - lo + 1;
            if (pb_used + nb + (long long)(s + 2) * 12 > pb_cap || pb_used + nb > 0x7fffff00ll) { end_reason = 3; break; }
            code_base = pb_used; pb_used += nb;
            if (tid == 0) { int* d = ws + ws_stride - 3ll * (s + 1); d[0] = lo; d[1] = hi; d[2] = (int)code_base; }
          }
        }
        const int o_m = data - base;
        const int o_i1 = o_m + width, o_d1 = o_m + 2 * width, o_i2 = o_m + 3 * width, o_d2 = o_m + 4 * width;
        int tmin[NCOMP], tmax[NCOMP];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) { tmin[c] = INT_MAX; tmax[c] = INT_MIN; }
        for (int k = lo + tid; k <= hi; k += T) {
          const int mo1_lo = mo1.get(ws, k - 1), i1e_lo = i1e.get(ws, k - 1), mo1_hi = mo1.get(ws, k + 1), d1e_hi = d1e.get(ws, k + 1);
          const int ins1 = max(mo1_lo, i1e_lo) + 1;
          const int del1 = max(mo1_hi, d1e_hi);
          int ins = ins1, del = del1;
          int ins2v = WFA_OFFSET_NULL, del2v = WFA_OFFSET_NULL;   // (2p) the second gap piece
          int pbcode = 0;
          if (has_i1) {
            ws[o_i1 + k] = ins1;
            if ((uint32_t)ins1 <= (uint32_t)tlen && (uint32_t)(ins1 - k) <= (uint32_t)plen) { tmin[NCOMP > 1 ? 1 : 0] = min(tmin[NCOMP > 1 ? 1 : 0], k); tmax[NCOMP > 1 ? 1 : 0] = max(tmax[NCOMP > 1 ? 1 : 0], k); }
          }
          if (has_d1) {
            ws[o_d1 + k] = del1;
            if ((uint32_t)del1 <= (uint32_t)tlen && (uint32_t)(del1 - k) <= (uint32_t)plen) { tmin[NCOMP > 2 ? 2 : 0] = min(tmin[NCOMP > 2 ? 2 : 0], k); tmax[NCOMP > 2 ? 2 : 0] = max(tmax[NCOMP > 2 ? 2 : 0], k); }
          }
          if (NCOMP == 5) {
            const int mo2_lo = mo2.get(ws, k - 1), i2e_lo = i2e.get(ws, k - 1), mo2_hi = mo2.get(ws, k + 1), d2e_hi = d2e.get(ws, k + 1);
            const int ins2 = max(mo2_lo, i2e_lo) + 1;
            const int del2 = max(mo2_hi, d2e_hi);
            if (PB) pbcode = ((i1e_lo >= mo1_lo) ? 8 : 0) | ((d1e_hi >= mo1_hi) ? 16 : 0) | ((i2e_lo >= mo2_lo) ? 32 : 0) | ((d2e_hi >= mo2_hi) ? 64 : 0);
            if (has_i2) {
              ws[o_i2 + k] = ins2;
              if ((uint32_t)ins2 <= (uint32_t)tlen && (uint32_t)(ins2 - k) <= (uint32_t)plen) { tmin[NCOMP - 2] = min(tmin[NCOMP - 2], k); tmax[NCOMP - 2] = max(tmax[NCOMP - 2], k); }
            }
            if (has_d2) {
              ws[o_d2 + k] = del2;
              if ((uint32_t)del2 <= (uint32_t)tlen && (uint32_t)(del2 - k) <= (uint32_t)plen) { tmin[NCOMP - 1] = min(tmin[NCOMP - 1], k); tmax[NCOMP - 1] = max(tmax[NCOMP - 1], k); }
            }
            ins = max(ins1, ins2);
            del = max(del1, del2);
            ins2v = ins2; del2v = del2;
          }
          const int x1 = mx.get(ws, k) + 1;
          int mv = max(del, max(x1, ins));
          if (PB) {
            // the choice the backtrace would make on equal offsets (R/wavefront_backtrace.c:49-59): mismatch > D2 > D1 > I2 > I1,
            // extension > opening
            if (NCOMP == 5) {
              const int best = mv;
              const int mc = (x1 >= best) ? 0 : (del2v >= best) ? 2 : (del1 >= best) ? 1 : (ins2v >= best) ? 4 : 3;
              pbcode |= mc;
            } else {
              const int mc = (x1 >= max(del1, ins1)) ? 0 : ((del1 >= ins1) ? 1 : 2);
              pbcode = mc | ((i1e_lo >= mo1_lo) ? 4 : 0) | ((d1e_hi >= mo1_hi) ? 8 : 0);
            }
            pb_codes[code_base + (k - lo)] = (uint8_t)pbcode;
          }
          // only M is clamped (R/wavefront_compute_affine.c:80-84)
          if ((uint32_t)mv > (uint32_t)tlen || (uint32_t)(mv - k) > (uint32_t)plen) mv = WFA_OFFSET_NULL;
          else { tmin[0] = min(tmin[0], k); tmax[0] = max(tmax[0], k); }
          ws[o_m + k] = mv;
        }
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const int mn = wave_min(tmin[c]), mxk = wave_max(tmax[c]);
          if ((tid & 63) == 0) {
            if (mn != INT_MAX) atomicMin(&TR[c], mn);
            if (mxk != INT_MIN) atomicMax(&TR[NCOMP + c], mxk);
          }
        }
        __syncthreads();
        // trimmed limits (R/wavefront_compute.c:571-605): first/last in-bounds offset; none -> null
        int tlo[NCOMP], thi[NCOMP];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const int mn = TR[c], mxk = TR[NCOMP + c];
          const bool has = (c == 0) || (c == 1 && has_i1) || (c == 2 && has_d1) ||
                           (NCOMP == 5 && c == 3 && has_i2) || (NCOMP == 5 && c == 4 && has_d2);
          if (has && mn != INT_MAX) { tlo[c] = mn; thi[c] = mxk; } else { tlo[c] = 1; thi[c] = -1; }
        }
        if (!FULL && ef_seed && s % (-cfg.match) == 0) {
          // wavefront_compute_endsfree_init (R/wavefront_compute.c:171-213) on the computed range [lo, hi], before the trim: a
          // begin-free cell replaces what compute-next put on its diagonal unless that is further along; a cell beyond the range
          // extends it (NULL in between).  The cells are in bounds, so the trimmed range grows to hold them.
          const int ek = s / (-cfg.match);
          auto grow = [&](int k) { if (tlo[0] > thi[0]) { tlo[0] = k; thi[0] = k; } else { tlo[0] = min(tlo[0], k); thi[0] = max(thi[0], k); } };
          if (cfg.tbf >= ek) {
            if (hi >= ek) {
              // (ek < lo: the reference compares with a cell outside the wavefront; whatever it writes stays outside)
              if (ek >= lo && ws[o_m + ek] <= ek) { if (tid == 0) ws[o_m + ek] = ek; grow(ek); }
            } else {
              for (int k = hi + 1 + tid; k <= ek; k += T) ws[o_m + k] = (k == ek) ? ek : WFA_OFFSET_NULL;
              grow(ek);
            }
          }
          if (cfg.pbf >= ek) {
            if (lo <= -ek) {
              if (-ek <= hi && ws[o_m - ek] <= 0) { if (tid == 0) ws[o_m - ek] = 0; grow(-ek); }
            } else {
              for (int k = -ek + tid; k < lo; k += T) ws[o_m + k] = (k == -ek) ? 0 : WFA_OFFSET_NULL;
              grow(-ek);
            }
          }
        }
        cur_exists = 1; cur_lo = tlo[0]; cur_hi = thi[0]; cur_idx0 = o_m;
        if (tid == 0) {
#pragma unroll
          for (int c = 0; c < NCOMP; ++c) { mslot[MT::LO + c] = tlo[c]; mslot[MT::HI + c] = thi[c]; }
          mslot[MT::BASE] = base; mslot[MT::WIDTH] = width; mslot[MT::DATA] = data; mslot[MT::EXISTS] = 1;
        }
      }
      if (s >= cfg.max_steps) { end_reason = 4; break; }  // R/wavefront_unialign.c:102-107
    }

    // =============================== finish ===============================
    if (tid == 0) {
      int out_score, out_status;
      long long cbeg = FULL ? a.cigar_off[pair + 1] : 0;
      int clen = 0;
      if (end_reason == 3) {
        if (FULL && a.fb_list) {
          // arena too small: hand the pair back to the host, which re-runs it with a larger arena
          out_status = WFA_INTERNAL_OVERFLOW; out_score = 0;
          a.fb_list[atomicAdd(a.fb_count, 1u)] = pair;
        } else {
          out_status = -200; out_score = INT_MIN;  // WF_STATUS_OOM
        }
      } else if (end_reason == 4) {
        out_status = -100; out_score = -cfg.max_steps;
      } else if (!FULL) {
        if (end_reason == 1) { out_score = classic_score(cfg, plen, tlen, s); out_status = 0; }
        else {
          // the reference evaluates the score at its unset end position (k=INT_MAX, offset=NULL)
          out_score = (cfg.metric <= 1) ? s : (cfg.match == 0) ? -s : (int)(((long long)(-cfg.match) * 1 - s) / 2);
          out_status = 1;
        }
      } else if (PB) {
        if (end_reason == 1) {
          // ---- walk the origin codes back from the end cell (R/wavefront_backtrace.c:320-529 with the choices made at
          // compute time); events go behind the code bytes, an event flagged 0x80 lands in M (a run of matches follows it)
          uint8_t* const ev = pb_codes + pb_used;
          const long long ev_cap = pb_cap - pb_used - (long long)(s + 2) * 12;
          const int sx = cfg.x, so1 = cfg.o1 + cfg.e1, se1 = cfg.e1, so2 = cfg.o2 + cfg.e2, se2 = cfg.e2;
          int sc = s, k = end_k, comp = 0;
          long long nev = 0;
          bool pb_fail = false;
          while (sc > 0) {
            if (nev >= ev_cap) { pb_fail = true; break; }
            const int* d = ws + ws_stride - 3ll * (sc + 1);
            const int cd = (k >= d[0] && k <= d[1]) ? pb_codes[(long long)d[2] + (k - d[0])] : 0;
            const uint8_t flag = (comp == 0) ? 0x80 : 0;
            int src;   // 0 mismatch, 1 D1, 2 D2, 3 I1, 4 I2
            if (NCOMP == 5) src = (comp == 0) ? (cd & 7) : (comp == 1) ? 3 : (comp == 2) ? 1 : (comp == 3) ? 4 : 2;
            else src = (comp == 0) ? ((cd & 3) == 0 ? 0 : ((cd & 3) == 1 ? 1 : 3)) : (comp == 1 ? 3 : 1);
            const int bi1 = (NCOMP == 5) ? 8 : 4, bd1 = (NCOMP == 5) ? 16 : 8;
            if (src == 0) { ev[nev++] = (uint8_t)('X' | 0x80); sc -= sx; }
            else if (src == 1) { ev[nev++] = (uint8_t)('D' | flag); ++k; if (cd & bd1) { sc -= se1; comp = 2; } else { sc -= so1; comp = 0; } }
            else if (src == 2) { ev[nev++] = (uint8_t)('D' | flag); ++k; if (cd & 64) { sc -= se2; comp = 4; } else { sc -= so2; comp = 0; } }
            else if (src == 3) { ev[nev++] = (uint8_t)('I' | flag); --k; if (cd & bi1) { sc -= se1; comp = 1; } else { sc -= so1; comp = 0; } }
            else { ev[nev++] = (uint8_t)('I' | flag); --k; if (cd & 32) { sc -= se2; comp = 3; } else { sc -= so2; comp = 0; } }
          }
          if (pb_fail || sc < 0) {
            if (a.fb_list) { out_status = WFA_INTERNAL_OVERFLOW; out_score = 0; a.fb_list[atomicAdd(a.fb_count, 1u)] = pair; }
            else { out_status = -200; out_score = INT_MIN; }
          } else {
            // ---- unpack forwards from the cell of wavefront 0 on diagonal k (ends-free: offset max(k, 0), R/wavefront_aligner.c:259-302)
            uint8_t* const out = a.cigar_ops + a.cigar_off[pair];
            long long n = 0;
            auto emit = [&](char c, int cnt) { for (int i = 0; i < cnt; ++i) out[n++] = (uint8_t)c; };
            int h = max(k, 0), v = h - k;
            emit('I', h); emit('D', v);
            { const int e = seq.extend(h - v, h); emit('M', e - h); v += e - h; h = e; }
            for (long long e_ = nev - 1; e_ >= 0; --e_) {
              const int op = ev[e_] & 0x7F;
              if (op == 'X') { emit('X', 1); ++v; ++h; }
              else if (op == 'I') { emit('I', 1); ++h; }
              else { emit('D', 1); ++v; }
              if (ev[e_] & 0x80) { const int e = seq.extend(h - v, h); emit('M', e - h); v += e - h; h = e; }
            }
            emit('I', tlen - h); emit('D', plen - v);
            cbeg = a.cigar_off[pair];
            clen = (int)n;
            out_score = classic_score(cfg, end_off - end_k, end_off, s);
            out_status = 0;
          }
        } else {
          out_score = INT_MIN; out_status = 1;
        }
      } else {
        if (end_reason == 1) {
          OpsWriter ops;
          ops.buf = a.cigar_ops + a.cigar_off[pair];
          ops.begin = (long long)plen + tlen;
          if (NCOMP == 1) backtrace_linear(ws, ws_stride, cfg, plen, tlen, s, end_k, end_off, ops);
          else backtrace<NCOMP == 1 ? 3 : NCOMP>(ws, ws_stride, cfg, plen, tlen, s, end_k, end_off, ops);
          cbeg = a.cigar_off[pair] + ops.begin;
          clen = (int)((long long)plen + tlen - ops.begin);
          out_score = classic_score(cfg, end_off - end_k, end_off, s);
          out_status = 0;
        } else {
          out_score = INT_MIN; out_status = 1;  // empty CIGAR after maxtrim (R/alignment/cigar.c:473-613)
        }
      }
      if (FULL && cfg.biwfa_top && out_status != WFA_INTERNAL_OVERFLOW) {
        // standing in for BiWFA's top-level base case: the score is never written there, and whatever is not a completed
        // alignment (step limit, unreachable) comes back as "unattainable" with no ops
        out_score = INT_MIN;
        if (out_status != 0) { out_status = WFA_STATUS_UNATTAINABLE; cbeg = a.cigar_off[pair]; clen = 0; }
      }
      a.score[pair] = out_score;
      a.status[pair] = out_status;
      if (FULL) { a.cigar_begin[pair] = cbeg; a.cigar_len[pair] = clen; }
    }
  }
}

// host entry points, one translation unit per component count (csrc/k_general.hip, -DWFA_TU_INDEX = 0 / 1 / 2 for
// NCOMP = 1 / 3 / 5)
#ifndef __HIPCC_RTC__   // ---- host side ----
int launch_general_c1(bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream);
int launch_general_c3(bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream);
int launch_general_c5(bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream);

template <int NCOMP>
inline int launch_general_ncomp(bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream) {
  const size_t smem = ((size_t)a.cfg.scope * Meta<NCOMP>::INTS + 2 * NCOMP + 8) * sizeof(int);
  const dim3 g(grid), t(threads);
  if constexpr (NCOMP >= 3) {
    if (full && pb) {
      if (packed) hipLaunchKernelGGL((wfa_general_kernel<NCOMP, true, true, true>), g, t, smem, stream, a);
      else hipLaunchKernelGGL((wfa_general_kernel<NCOMP, false, true, true>), g, t, smem, stream, a);
      return hipGetLastError() == hipSuccess ? 0 : -1;
    }
  }
  if (packed) {
    if (full) hipLaunchKernelGGL((wfa_general_kernel<NCOMP, true, true>), g, t, smem, stream, a);
    else hipLaunchKernelGGL((wfa_general_kernel<NCOMP, true, false>), g, t, smem, stream, a);
  } else {
    if (full) hipLaunchKernelGGL((wfa_general_kernel<NCOMP, false, true>), g, t, smem, stream, a);
    else hipLaunchKernelGGL((wfa_general_kernel<NCOMP, false, false>), g, t, smem, stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// pb: the piggy-back history (full scope, gap-affine / gap-affine-2p; ignored otherwise)
inline int launch_general_any(int ncomp, bool packed, bool full, bool pb, const WfaKernelArgs& a, int grid, int threads, hipStream_t stream) {
  if (ncomp == 1) return launch_general_c1(packed, full, false, a, grid, threads, stream);
  if (ncomp == 3) return launch_general_c3(packed, full, pb, a, grid, threads, stream);
  return launch_general_c5(packed, full, pb, a, grid, threads, stream);
}

#endif  // __HIPCC_RTC__

}  // namespace wfa
