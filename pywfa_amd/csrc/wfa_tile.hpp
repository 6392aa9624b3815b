// wfa_tile.hpp — exact gap-affine / gap-affine-2p alignment of long reads with wide wavefronts, TEMPORALLY BLOCKED
// (VERDICT r03 item 2): the M / I / D wavefront rows live in the workgroup's slice of the HBM workspace as in wfa_wide.hpp's
// workspace form, but a WAVE advances a block of diagonals by T score steps inside an LDS tile (block + a halo of T diagonals
// per side), so a row is written once and read once per T steps instead of five (2p: nine) reads and three (five) writes per
// step — and the T steps of a tile need no workgroup barrier at all: one barrier per super-step.
//
// R = /root/reference/pywfa/WFA2_lib/wavefront.  Per cell and step exactly wfa_wide.hpp's pass:
//   compute-next  R/wavefront_compute_affine.c:44-86, R/wavefront_compute_affine2p.c:45-106 (only M clamped to the sequences)
//   extend        R/wavefront_extend_kernels.c:64-88 on the 2-bit codes (16-base first probe, then 32 bases per round)
//   termination   R/wavefront_termination.c:37-61 (end-to-end), :115-162 (ends-free: lowest k of the first step that ends)
//   limit         R/wavefront_unialign.c:102-107 (max_steps)
//   trimming      R/wavefront_compute.c:571-605 — not performed per step: csrc/wfa_tile_cell.hpp explains why the untrimmed
//                 superset computes the same values, and the check that hands a pair on (to wfa_wide.hpp's step-by-step form)
//                 when it would not.  tools/tile_model.cpp + tests/test_tile_model.py pin that argument on CPU.
// Penalties are run-time values (rings are indexed in LDS, nothing is unrolled over them): any x, o, e (, o2, e2), match = 0.
// Scope: 2-bit pairs, no heuristic, end-to-end / ends-free, score or full CIGAR (piggy-back codes + directory exactly as
// wfa_wide.hpp, walked by wide_walk_unpack), reads of up to 32 000 bases (int16 rows; round 6: NULL = -32768).
//
// Layout.  Column c = k + plen.  Block b owns columns [b Bw, (b+1) Bw), Bw = Wt - 2 T; its tile holds columns
// [b Bw - T, (b+1) Bw + T).  HBM row element = column + T (so tile b starts at element b Bw: dword-aligned).  HBM ring slots
// by tile_hbm_slot (rows a super-step reads and rows it writes never share a slot: the blocks of one super-step are
// independent, each wave takes every nwaves-th block).  LDS per wave: tile_lds_rows rows of Wt + 4 int16.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <type_traits>
#include "wfa_common.hpp"
#include "wfa_hip.h"
#include "wfa_tile_cell.hpp"
#include "wfa_wide.hpp"

namespace wfa {

struct TileArgs {
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
  int32_t* hist;          // full scope: slice of workgroup b = hist + b * hist_stride (ints): codes bottom-up, directory top-down
  long long hist_stride;
  short* rows;            // slice of workgroup b = rows + b * rows_stride (int16 elements)
  long long rows_stride;
  int rwh;                // elements per HBM row (even)
  TileGeom g;
  int gs;                 // score per step
  int ef, pbf, pef, tbf, tef;
  int max_steps;
  int seq_words;          // LDS words per sequence (>= words of the longest sequence + 3)
  uint32_t* dbg;          // development aid (WFA_HIP_STAGE_TIMING): [0] super-steps, [1] careful passes, [2] tiles; nullptr otherwise
};

#define WFA_TILE_CTRL_INTS 16
#define WFA_TILE_MAX_T 32
#define WFA_TILE_MAX_ROWS 64    // rows a tile loads / writes back: DM + 2 E + 2 E2 candidates
// ctrl: [0] end key, [1] taint, [2] violation, [8 + comp] LDS ring position of row t0 of component comp
static inline size_t tile_smem_bytes(const TileGeom& g, int seq_words, int nwaves, int cell_bytes = 2) {
  return (size_t)(WFA_TILE_CTRL_INTS + 3 * WFA_TILE_MAX_T + 16 * WFA_TILE_MAX_T + 4 * WFA_TILE_MAX_ROWS) * 4 + (size_t)2 * seq_words * 4 +
         (size_t)nwaves * tile_lds_rows(g) * tile_lds_pitch(g) * cell_bytes;
}
// candidate row i of the load / write-back tables -> (component, distance d >= 1)
WFA_TILE_HD int tile_cand_count(const TileGeom& g) { return g.DM + 2 * g.E + (g.OE2 > 0 ? 2 * g.E2 : 0); }
WFA_TILE_HD void tile_cand(const TileGeom& g, int i, int* comp, int* d) {
  if (i < g.DM) { *comp = 0; *d = i + 1; return; }
  i -= g.DM;
  if (i < g.E) { *comp = 1; *d = i + 1; return; }
  i -= g.E;
  if (i < g.E) { *comp = 2; *d = i + 1; return; }
  i -= g.E;
  if (i < g.E2) { *comp = 3; *d = i + 1; return; }
  *comp = 4; *d = i - g.E2 + 1;
}

__device__ __forceinline__ uint32_t tile_ffbl(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }

// NCH: 64-column chunks of a tile (Wt = 64 NCH): the chunks of a step run interleaved, so their LDS latencies overlap
// W32 (round 6): int32 cells in the rows and the tiles — reads beyond 32 000 bases (exact 100 kb), which ran the step-by-step
// wfa_wide_kernel<.., W32> until now.  Same schedule; a cell is a dword (CB = 4 bytes, one per dword instead of two), NULL is
// WFA_OFFSET_NULL, the end key is 64 bits (step << 32 | diagonal + 2^30).
template <bool FULL, bool TWO, int NCH, bool W32 = false>
__global__ void __launch_bounds__(512)
wfa_tile_kernel(const TileArgs a) {
  constexpr int NC = TWO ? 5 : 3;
  constexpr int CB = W32 ? 4 : 2;                       // bytes per cell
  constexpr int CS = W32 ? 0 : 1;                       // cells -> dwords: >> CS
  constexpr int NULLV = W32 ? (int)0xC0000000 : WFA_TILE_NULL;
  constexpr uint32_t NULLDW = W32 ? 0xC0000000u : WFA_TILE_NULL2;
  typedef typename std::conditional<W32, int, short>::type cell_t;
  extern __shared__ int tsm[];
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: everything per tile then lives in scalar registers)
  // (built from scalars: a struct copied out of the kernel arguments and then selected from by component ends up in scratch)
  const TileGeom g = {a.g.X, a.g.OE, a.g.E, TWO ? a.g.OE2 : 0, TWO ? a.g.E2 : 0, a.g.T, a.g.Wt, a.g.DM};
  const int T = g.T, Wt = g.Wt, Bw = Wt - 2 * T;
  const int pitch = tile_lds_pitch(g);            // halfs
  int* const ctrl = tsm;
  int* const lo_j = tsm + WFA_TILE_CTRL_INTS;
  int* const hi_j = lo_j + WFA_TILE_MAX_T;
  int* const base_j = hi_j + WFA_TILE_MAX_T;
  int* const stats = base_j + WFA_TILE_MAX_T;      // careful pass: [j][gap comp 0..3][in_min, in_max, pe_min, pe_max]
  // per super-step tables of the rows a tile loads (row t0 - d) and writes back (row t0 + T - d): LDS row, HBM row (-1: fill
  // with NULL, -2: not needed)
  int* const ld_lds = stats + 16 * WFA_TILE_MAX_T;
  int* const ld_hbm = ld_lds + WFA_TILE_MAX_ROWS;
  int* const wb_lds = ld_hbm + WFA_TILE_MAX_ROWS;
  int* const wb_hbm = wb_lds + WFA_TILE_MAX_ROWS;
  uint32_t* const sP = reinterpret_cast<uint32_t*>(wb_hbm + WFA_TILE_MAX_ROWS);
  const int ncand = tile_cand_count(g);
  uint32_t* const sT = sP + a.seq_words;
  cell_t* const tile = reinterpret_cast<cell_t*>(sT + a.seq_words) + (size_t)wave * tile_lds_rows(g) * pitch;
  cell_t* const rows = reinterpret_cast<cell_t*>(a.rows) + (long long)blockIdx.x * a.rows_stride;
  int* const hist = FULL ? a.hist + (long long)blockIdx.x * a.hist_stride : nullptr;
  uint8_t* const pb_codes = FULL ? reinterpret_cast<uint8_t*>(hist) : nullptr;
  const long long pb_cap = FULL ? a.hist_stride * 4 : 0;
  const uint32_t nwork = a.nwork_dev ? *a.nwork_dev : a.nwork;
  const int nhrows = tile_hbm_rows(g);

  // guard columns of this wave's tile: NULL, never written again
  for (int r = lane; r < tile_lds_rows(g); r += 64) {
    cell_t* row = tile + (size_t)r * pitch;
    row[0] = row[1] = row[pitch - 2] = row[pitch - 1] = (cell_t)NULLV;
  }

  for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t pair = a.worklist ? a.worklist[wi] : wi;
    const WfaPairMeta pm = a.meta[pair];
    const int plen = pm.plen, tlen = pm.tlen;
    const int ak = tlen - plen;
    const int pbf = a.ef ? a.pbf : 0, tbf = a.ef ? a.tbf : 0;
    __syncthreads();   // the previous pair is done with LDS and the rows
    // W32 (reads beyond 32 kb): the sequences stay in global memory (a.seq_words = 0: 2 x 6+ KB per pair would leave one workgroup per CU);
    // the probes read them through the vector cache — a probe never reads beyond a sequence's two look-ahead words, which exist
    const uint32_t* const gp = a.words + pm.p_woff;
    const uint32_t* const gt = a.words + pm.t_woff;
    const uint32_t* const qP = W32 ? gp : sP;
    const uint32_t* const qT = W32 ? gt : sT;
    {
      const int nwp = (plen + 15) >> 4, nwt = (tlen + 15) >> 4;
      if (!W32) for (int i = tid; i < a.seq_words; i += blockDim.x) { sP[i] = (i < nwp) ? gp[i] : 0u; sT[i] = (i < nwt) ? gt[i] : 0u; }
      // the rows this pair can touch start as NULL (blocks 0 .. nb-1 and their halos)
      const int nb = (plen + tlen + 1 + Bw - 1) / Bw;
      const int used32 = min(a.rwh, nb * Bw + 2 * T) >> CS;
      uint32_t* r32 = reinterpret_cast<uint32_t*>(rows);
      for (int r = 0; r < nhrows; ++r)
        for (int i = tid; i < used32; i += blockDim.x) r32[(size_t)r * (a.rwh >> CS) + i] = NULLDW;
      if (tid < WFA_TILE_CTRL_INTS) ctrl[tid] = (tid == 0 || (W32 && (tid == 4 || tid == 5))) ? -1 : 0;   // ([4..5]: W32's 64-bit end key)
    }
    int end_reason = ((!W32 && (plen > WFA_TILE_MAX_LEN || tlen > WFA_TILE_MAX_LEN)) || (plen + tlen + 1 + Bw - 1) / Bw * Bw + 2 * T > a.rwh) ? 3 : 0;   // 1 reached, 3 handed on, 4 step limit
    int end_k = 0, end_t = 0;
    long long pb_used = 0;
    const int t_lim = (a.max_steps == INT_MAX) ? INT_MAX : (int)max(1ll, ((long long)a.max_steps + a.gs - 1) / a.gs);   // first step whose score reaches the limit
    __syncthreads();

    for (int ss = 0; !end_reason; ++ss) {
      const int t0 = ss * T;
      if (!W32 && t0 + T > WFA_TILE_MAX_LEN) { end_reason = 3; break; }   // (int16 rows: a dead value gains at most 1 per step and must stay negative)
      // ---- the T steps' diagonal ranges and, full scope, their code bytes + directory records ----
      if (tid < T) { lo_j[tid] = tile_lo(g, t0 + tid, plen, pbf); hi_j[tid] = tile_hi(g, t0 + tid, tlen, tbf); }
      for (int i = tid; i < ncand; i += blockDim.x) {
        int comp, d;
        tile_cand(g, i, &comp, &d);
        // (dword offsets: of the row's column 0 / first own column inside the wave's tile, of the row inside the workgroup's rows)
        ld_lds[i] = (((comp == 0) ? tile_lds_slot_m_old(g, t0, d) : tile_lds_slot(g, comp, t0 - d)) * pitch + 2) >> CS;
        ld_hbm[i] = !tile_loads_row(g, comp, d) ? -2 : (t0 - d < 0) ? -1 : tile_hbm_slot(g, comp, t0 - d) * (a.rwh >> CS);
        wb_lds[i] = (tile_lds_slot(g, comp, t0 + T - d) * pitch + 2 + T) >> CS;
        wb_hbm[i] = (d <= T) ? (tile_hbm_slot(g, comp, t0 + T - d) * a.rwh + T) >> CS : -2;
      }
      if (tid < NC) ctrl[8 + tid] = t0 % tile_lds_ring_depth(g, tid);
      long long need = 0;
      if (FULL) {
        for (int j = 0; j < T; ++j) need += (long long)(tile_hi(g, t0 + j, tlen, tbf) - tile_lo(g, t0 + j, plen, pbf) + 1);
        if (pb_used + need + (long long)(t0 + T + 2) * 12 + 64 > pb_cap || pb_used + need > 0x7fffff00ll) { end_reason = 3; break; }
        if (tid == 0) {
          long long bse = pb_used;
          for (int j = 0; j < T; ++j) {
            const int l = tile_lo(g, t0 + j, plen, pbf), h = tile_hi(g, t0 + j, tlen, tbf);
            base_j[j] = (int)bse;
            int* d = hist + a.hist_stride - 3ll * (t0 + j + 1);
            d[0] = l; d[1] = h; d[2] = (int)bse;
            bse += h - l + 1;
          }
        }
      }
      __syncthreads();
      const int b_first = (lo_j[T - 1] + plen) / Bw, b_last = (hi_j[T - 1] + plen) / Bw;
      // the row tables, one entry per lane (read with v_readlane inside the tile loops: no LDS round trip per row)
      const int my_ld_lds = (lane < ncand) ? ld_lds[lane] : 0, my_ld_hbm = (lane < ncand) ? ld_hbm[lane] : -2;
      const int my_wb_lds = (lane < ncand) ? wb_lds[lane] : 0, my_wb_hbm = (lane < ncand) ? wb_hbm[lane] : -2;
      const unsigned long long ld_mask = __ballot(my_ld_hbm >= 0), null_mask = __ballot(my_ld_hbm == -1), wb_mask = __ballot(my_wb_hbm >= 0);
      uint32_t* const tile32 = reinterpret_cast<uint32_t*>(tile);
      uint32_t* const rows32 = reinterpret_cast<uint32_t*>(rows);

      for (int pass = 0; pass < 2; ++pass) {
        const bool careful = pass == 1;
        if (careful) {
          for (int i = tid; i < 16 * T; i += blockDim.x) stats[i] = (i & 1) ? INT_MIN : INT_MAX;   // in_min, in_max, pe_min, pe_max
          __syncthreads();
        }
        bool taint = false;
        // byte offsets (from this lane's column 0 of LDS row 0) of the rows a step reads and writes; they advance by one row per
        // step and wrap inside their ring: 0 X, 1 O, 2 I, 3 D, 4 O2, 5 I2, 6 D2 (inputs), 7 M, 8 I, 9 D, 10 I2, 11 D2 (outputs)
        const int pitch_b = CB * pitch;
        constexpr int NP = TWO ? 12 : 10;
        int pos0[12], rlo[12], rhi[12];
        {
          const int nmr = tile_m_ring(g);
          const int rbI = nmr, rbD = rbI + g.E + 1, rbI2 = rbD + g.E + 1, rbD2 = rbI2 + g.E2 + 1;
          const int ring_base[5] = {0, rbI, rbD, rbI2, rbD2};
          const int ring_dep[5] = {nmr, g.E + 1, g.E + 1, g.E2 + 1, g.E2 + 1};
          const int comp_of[12] = {0, 0, 1, 2, 0, 3, 4, 0, 1, 2, 3, 4};
          const int lag_of[12] = {g.X, g.OE, g.E, g.E, g.OE2, g.E2, g.E2, 0, 0, 0, 0, 0};
#pragma unroll
          for (int i = 0; i < 12; ++i) {
            if (!TWO && (i == 4 || i == 5 || i == 6 || i >= 10)) { pos0[i] = rlo[i] = rhi[i] = 0; continue; }
            if (TWO && i == 4 && tile_far(g)) {   // M at lag OE2 from the far buffer: row j for step j
              rlo[i] = tile_lds_far_base(g) * pitch_b; rhi[i] = rlo[i] + (T + 1) * pitch_b; pos0[i] = rlo[i];
              continue;
            }
            const int cmp = comp_of[i];
            int p = __builtin_amdgcn_readfirstlane(ctrl[8 + cmp]) - lag_of[i];
            if (p < 0) p += ring_dep[cmp];
            rlo[i] = ring_base[cmp] * pitch_b;
            rhi[i] = (ring_base[cmp] + ring_dep[cmp]) * pitch_b;
            pos0[i] = rlo[i] + p * pitch_b;
          }
        }
#ifdef WFA_TILE_PROFILE
        long long pt_load = 0, pt_comp = 0, pt_wb = 0;
        unsigned pn_long = 0, pn_end = 0;
#endif
        for (int b = b_first + wave; b <= b_last; b += nwaves) {
#ifdef WFA_TILE_PROFILE
          const long long pc0 = clock64();
#endif
          // ---- tile load: the rows the T steps read, columns [b Bw - T, (b+1) Bw + T) (the tables sit one entry per lane) ----
          const uint32_t vcol = (uint32_t)(b * (Bw >> CS) + lane);    // this lane's dword of a row
          for (unsigned long long m = ld_mask; m; m &= m - 1) {
            const int i = (int)__builtin_ctzll(m);
            uint32_t* const dst = tile32 + __builtin_amdgcn_readlane(my_ld_lds, i);
            const uint32_t* const src = rows32 + ((uint32_t)__builtin_amdgcn_readlane(my_ld_hbm, i) + vcol);
#pragma unroll
            for (int c0 = 0; c0 < ((64 * NCH) >> CS); c0 += 64)
              if (c0 + lane < ((64 * NCH) >> CS)) __builtin_amdgcn_global_load_lds(src + c0, dst + c0, 4, 0, 0);
          }
          for (unsigned long long m = null_mask; m; m &= m - 1) {   // rows before score 0 (the first super-steps only)
            uint32_t* const dst = tile32 + __builtin_amdgcn_readlane(my_ld_lds, (int)__builtin_ctzll(m));
            for (int c = lane; c < (Wt >> CS); c += 64) dst[c] = NULLDW;
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the rows are in LDS
          __builtin_amdgcn_wave_barrier();
#ifdef WFA_TILE_PROFILE
          const long long pc1 = clock64();
#endif
          // ---- T steps inside the tile.  Per lane and chunk (column = 64 c + lane) the diagonal and its limits do not change
          // with the step; what depends on "the block owns this column" or "the diagonal holds cells" is folded into per-chunk
          // constants so that a step tests nothing but offsets ----
          int kk[NCH], bas[NCH], span[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int col = 64 * c + lane;
            kk[c] = b * Bw - T + col - plen;
            const int l = min(tlen, plen + kk[c]);     // the largest offset on the diagonal
            const int s0 = max(kk[c], 0);              // the smallest
            const bool kin = l >= s0;                  // (a diagonal outside [-plen, tlen] holds no cell: always trimmed)
            bas[c] = kin ? s0 : INT_MAX;               // in bounds <=> (unsigned)(offset - bas) <= span
            span[c] = kin ? l - s0 : 0;
          }
          // the block's own columns: all but the first T of chunk 0 and the last T of the last chunk (T <= 32)
          const int kown = b * Bw - plen;              // own diagonals: kown .. kown + Bw - 1
          const int span_first = (lane >= T) ? span[0] : INT_MAX / 2;              // taint test of the edge chunks: never on a halo column
          const int span_last = (lane < 64 - T) ? span[NCH - 1] : INT_MAX / 2;
          int tmax = INT_MIN;
          int pos[12];
#pragma unroll
          for (int i = 0; i < 12; ++i) pos[i] = pos0[i];
          const char* const tb = reinterpret_cast<const char*>(tile) + 2 * CB + CB * lane;   // this lane's column 0 (chunk 0) of LDS row 0
          auto step = [&](const int j, auto first_tag, auto careful_tag) {
            constexpr bool FIRST = decltype(first_tag)::value, CAREFUL = decltype(careful_tag)::value;
            const int t = t0 + j;
            const char* const pX = tb + pos[0];
            const char* const pO = tb + pos[1] - CB;        // (column - 1; column + 1 is two cells on)
            const char* const pI = tb + pos[2] - CB;
            const char* const pD = tb + pos[3] + CB;
            const char* const pO2 = tb + pos[4] - CB;
            const char* const pI2 = tb + pos[5] - CB;
            const char* const pD2 = tb + pos[6] + CB;
            auto ld = [](const char* p, int c) { return (int)*reinterpret_cast<const cell_t*>(p + 64 * CB * c); };
            auto st = [&](int i, int c, int v) { *reinterpret_cast<cell_t*>(const_cast<char*>(tb) + pos[i] + 64 * CB * c) = (cell_t)v; };
            // -- compute-next of every chunk (R/wavefront_compute_affine.c:44-86, R/wavefront_compute_affine2p.c:45-106) --
            int vm[NCH], code[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const int mx = ld(pX, c), mo_lo = ld(pO, c), mo_hi = ld(pO + 2 * CB, c), ie_lo = ld(pI, c), de_hi = ld(pD, c);
              int mo2_lo = NULLV, mo2_hi = NULLV, i2e_lo = NULLV, d2e_hi = NULLV;
              if (TWO) { mo2_lo = ld(pO2, c); mo2_hi = ld(pO2 + 2 * CB, c); i2e_lo = ld(pI2, c); d2e_hi = ld(pD2, c); }
              TileCell cc = tile_cell<TWO, FULL>(mx, mo_lo, mo_hi, ie_lo, de_hi, mo2_lo, mo2_hi, i2e_lo, d2e_hi);
              if (FIRST) { if (kk[c] >= -pbf && kk[c] <= tbf) cc.m_raw = max(kk[c], 0); }   // R/wavefront_aligner.c:251-310
              // taint: the untrimmed M candidate passes the end of an own diagonal (signed: a dead value is far below)
              tmax = max(tmax, (cc.m_raw - bas[c]) - ((NCH == 1) ? min(span_first, span_last) : (c == 0) ? span_first : (c == NCH - 1) ? span_last : span[c]));
              vm[c] = cc.m_raw; code[c] = cc.code;
              // (gap values on a diagonal without cells are stored as they come: they can only feed diagonals further out and M there is NULL)
              st(8, c, cc.i1); st(9, c, cc.d1);
              if (TWO) { st(10, c, cc.i2); st(11, c, cc.d2); }
              if (CAREFUL) {
                const int gv[4] = {cc.i1, cc.d1, cc.i2, cc.d2};
#pragma unroll
                for (int q = 0; q < NC - 1; ++q) {
                  const bool ok = (uint32_t)(kk[c] - kown) < (uint32_t)Bw && bas[c] != INT_MAX;   // own column of a diagonal with cells
                  const unsigned long long bi = __ballot(ok && (uint32_t)(gv[q] - bas[c]) <= (uint32_t)span[c]);
                  const unsigned long long bp = __ballot(ok && gv[q] - bas[c] > span[c]);
                  if (lane == 0) {
                    int* sp = stats + (j * 4 + q) * 4;
                    const int k0 = b * Bw - T + 64 * c - plen;
                    if (bi) { atomicMin(&sp[0], k0 + (int)__builtin_ctzll(bi)); atomicMax(&sp[1], k0 + 63 - (int)__builtin_clzll(bi)); }
                    if (bp) { atomicMin(&sp[2], k0 + (int)__builtin_ctzll(bp)); atomicMax(&sp[3], k0 + 63 - (int)__builtin_clzll(bp)); }
                  }
                }
              }
            }
            // -- extend M (R/wavefront_extend_kernels.c:64-88): a 16-base first probe of every chunk together, then 32-base rounds
            //    while any lane of any chunk still runs; never past either sequence end --
            int hh[NCH], left[NCH], rem = 0;
            bool m_in[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              m_in[c] = (uint32_t)(vm[c] - bas[c]) <= (uint32_t)span[c];
              hh[c] = m_in[c] ? vm[c] : max(kk[c], 0);   // (a lane without a cell probes some cell of its column and discards the result)
              left[c] = m_in[c] ? bas[c] + span[c] - vm[c] : 0;
              const int v = hh[c] - kk[c], h = hh[c];
              // (W32: a lane without a cell must not read beyond the sequences' look-ahead words: global memory, not a staged LDS area)
              const int pi = W32 ? min(v >> 4, (plen + 15) >> 4) : v >> 4, ti = W32 ? min(h >> 4, (tlen + 15) >> 4) : h >> 4;
              const uint32_t x = __builtin_amdgcn_alignbit(qP[pi + 1], qP[pi], (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(qT[ti + 1], qT[ti], (uint32_t)h << 1);
              const int m = min((int)(tile_ffbl(x) >> 1), min(16, left[c]));
              hh[c] += m; left[c] -= m;
              rem |= (m == 16) ? left[c] : 0;
            }
            if (__any(rem > 0)) {
#ifdef WFA_TILE_PROFILE
              ++pn_long;
#endif
              bool more[NCH];
#pragma unroll
              for (int c = 0; c < NCH; ++c) more[c] = m_in[c] && left[c] > 0 && ((hh[c] - vm[c]) == 16);
              bool any_more = true;
              while (__any(any_more)) {
                any_more = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                  if (more[c]) {
                    const int v = hh[c] - kk[c], h = hh[c];
                    const int pi = v >> 4, ti = h >> 4;
                    const uint32_t p0 = qP[pi], p1 = qP[pi + 1], p2 = qP[pi + 2], t0_ = qT[ti], t1 = qT[ti + 1], t2 = qT[ti + 2];
                    const uint32_t xl = __builtin_amdgcn_alignbit(p1, p0, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t1, t0_, (uint32_t)h << 1);
                    const uint32_t xh = __builtin_amdgcn_alignbit(p2, p1, (uint32_t)v << 1) ^ __builtin_amdgcn_alignbit(t2, t1, (uint32_t)h << 1);
                    const uint32_t fb = min(tile_ffbl(xl), tile_ffbl(xh) | 32u);
                    const int m = min((int)(fb >> 1), min(32, left[c]));
                    hh[c] += m; left[c] -= m;
                    more[c] = (m == 32) && (left[c] > 0);
                    any_more = any_more || more[c];
                  }
                }
              }
            }
            // -- the extended offsets: stores, termination --
            const int dhi = min(hi_j[j], kown + Bw - 1);                   // the step's diagonals this block owns: dlo .. dhi
            const int dlo = (dhi >= max(lo_j[j], kown)) ? max(lo_j[j], kown) : INT_MAX, dwd = (dlo != INT_MAX) ? dhi - dlo : 0;   // (none: no lane passes)
            uint8_t* const cj = FULL ? pb_codes + (long long)base_j[j] - lo_j[j] : nullptr;   // the code of diagonal k goes to cj[k]
            int reach = INT_MIN;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const int mfin = m_in[c] ? hh[c] : NULLV;   // only M is clamped (R/wavefront_compute_affine.c:80-84)
              st(7, c, mfin);
              hh[c] = mfin;
              reach = max(reach, a.ef ? mfin + max(0, -kk[c]) : mfin);
              if (FULL) { if ((uint32_t)(kk[c] - dlo) <= (uint32_t)dwd) cj[kk[c]] = (uint8_t)code[c]; }
            }
            // a cell can end the alignment only by reaching the end of the text (offset tlen) or of the pattern (offset - k = plen):
            // rare until the last steps, so one cheap test per step guards the exact ones
            if (__any(a.ef ? (reach >= min(tlen, plen)) : (reach >= tlen))) {
#ifdef WFA_TILE_PROFILE
              ++pn_end;
#endif
#pragma unroll
              for (int c = 0; c < NCH; ++c) {
                const int h = hh[c], v = h - kk[c];
                const bool own = (uint32_t)(kk[c] - kown) < (uint32_t)Bw;
                const bool fin = h >= 0 && own &&
                                 (a.ef ? ((h >= tlen && plen - v <= a.pef) || (v >= plen && tlen - h <= a.tef)) : (kk[c] == ak && h >= tlen));
                if (fin) {
                  if (W32) atomicMin(reinterpret_cast<unsigned long long*>(&ctrl[4]), ((unsigned long long)(unsigned)t << 32) | (unsigned)(kk[c] + (1 << 30)));
                  else atomicMin(reinterpret_cast<unsigned*>(&ctrl[0]), ((unsigned)t << 16) | (unsigned)(kk[c] + 32768));
                }
              }
            }
            // the rings advance
#pragma unroll
            for (int i = 0; i < NP; ++i) {
              if (!TWO && (i == 4 || i == 5 || i == 6)) continue;
              pos[i] += pitch_b;
              if (pos[i] == rhi[i]) pos[i] = rlo[i];
            }
            __builtin_amdgcn_wave_barrier();
          };
          {
            int j = 0;
            if (t0 == 0) {
              if (careful) step(0, std::true_type{}, std::true_type{}); else step(0, std::true_type{}, std::false_type{});
              j = 1;
            }
            if (careful) { for (; j < T; ++j) step(j, std::false_type{}, std::true_type{}); }
            else {
#pragma unroll 1
              for (; j < T; ++j) step(j, std::false_type{}, std::false_type{});
            }
          }
          taint = taint || tmax > 0;
#ifdef WFA_TILE_PROFILE
          const long long pc2 = clock64();
#endif
          // ---- write-back: the block's own columns of the rows later super-steps read (two rows in flight) ----
          {
            unsigned long long m = wb_mask;
            while (m) {
              const int i0 = (int)__builtin_ctzll(m); m &= m - 1;
              const bool two_rows = m != 0;
              const int i1 = two_rows ? (int)__builtin_ctzll(m) : i0;
              if (two_rows) m &= m - 1;
              const uint32_t* s0 = tile32 + __builtin_amdgcn_readlane(my_wb_lds, i0);
              const uint32_t* s1 = tile32 + __builtin_amdgcn_readlane(my_wb_lds, i1);
              uint32_t* d0 = rows32 + ((uint32_t)__builtin_amdgcn_readlane(my_wb_hbm, i0) + vcol);
              uint32_t* d1 = rows32 + ((uint32_t)__builtin_amdgcn_readlane(my_wb_hbm, i1) + vcol);
#pragma unroll
              for (int c0 = 0; c0 < ((64 * NCH) >> CS); c0 += 64) {
                if (c0 + lane < (Bw >> CS)) {
                  const uint32_t v0 = s0[c0 + lane], v1 = s1[c0 + lane];
                  d0[c0] = v0;
                  if (two_rows) d1[c0] = v1;
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
#ifdef WFA_TILE_PROFILE
          { const long long pc3 = clock64(); pt_load += pc1 - pc0; pt_comp += pc2 - pc1; pt_wb += pc3 - pc2; }
#endif
        }
#ifdef WFA_TILE_PROFILE
        {
          const long long pcb = clock64();
          __syncthreads();
          const long long pce = clock64();
          if (a.dbg && lane == 0) { atomicAdd(a.dbg + 3, (uint32_t)(pt_load >> 10)); atomicAdd(a.dbg + 4, (uint32_t)(pt_comp >> 10)); atomicAdd(a.dbg + 5, (uint32_t)(pt_wb >> 10)); atomicAdd(a.dbg + 6, (uint32_t)((pce - pcb) >> 10) * 0u + pn_end); atomicAdd(a.dbg + 7, pn_long); }
        }
#endif
        if (__any(taint) && lane == 0) atomicOr(&ctrl[1], 1);
        __syncthreads();   // rows written, end key / taint / statistics visible
        if (a.dbg && tid == 0) { atomicAdd(a.dbg + (careful ? 1 : 0), 1u); atomicAdd(a.dbg + 2, (uint32_t)(b_last - b_first + 1)); }
        if (!careful && ctrl[1] != 0) continue;   // a cell past the end of its diagonal: repeat the super-step with the trimming statistics
        if (careful) {
          // a gap cell past the end outside [first, last in-bounds] of its row is where the reference's trimming changes a value
          const unsigned long long key64 = W32 ? *reinterpret_cast<const unsigned long long*>(&ctrl[4]) : 0ull;
          const unsigned key = (unsigned)ctrl[0];
          const bool ended = W32 ? (key64 != ~0ull) : (key != 0xFFFFFFFFu);
          const int j_end = ended ? (W32 ? (int)(key64 >> 32) : (int)(key >> 16)) - t0 : T - 1;   // (steps after the end do not count)
          for (int i = tid; i < 4 * T; i += blockDim.x) {
            const int* st = stats + i * 4;
            if ((i >> 2) <= j_end && st[2] != INT_MAX && (st[0] == INT_MAX || st[2] < st[0] || st[3] > st[1])) atomicOr(&ctrl[2], 1);
          }
          __syncthreads();
        }
        break;
      }
      if (ctrl[2] != 0) { end_reason = 3; }
      else {
        const unsigned long long key64 = W32 ? *reinterpret_cast<const unsigned long long*>(&ctrl[4]) : 0ull;
        const unsigned key = (unsigned)ctrl[0];
        const bool ended = W32 ? (key64 != ~0ull) : (key != 0xFFFFFFFFu);
        const int t_end = ended ? (W32 ? (int)(key64 >> 32) : (int)(key >> 16)) : INT_MAX;
        if (t_lim < t0 + T && t_lim <= t_end) end_reason = 4;
        else if (t_end != INT_MAX) { end_reason = 1; end_t = t_end; end_k = W32 ? (int)(unsigned)(key64 & 0xFFFFFFFFull) - (1 << 30) : (int)(key & 0xFFFFu) - 32768; }
      }
      if (FULL) { for (int j = 0; j < T; ++j) pb_used += (long long)(hi_j[j] - lo_j[j] + 1); }
      __syncthreads();   // everyone has read ctrl / the tables
      if (tid == 0) ctrl[1] = 0;
    }

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
        out_score = -(end_t * a.gs);
        if (FULL) {
          const long long n = wide_walk_unpack<TWO>(hist, a.hist_stride, pb_codes, pb_codes + pb_used, pb_cap - pb_used - (long long)(end_t + T + 2) * 12,
                                                    end_t, end_k, g.X, g.OE, g.E, g.OE2, g.E2, qP, qT, plen, tlen, a.cigar_ops + a.cigar_off[pair]);
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

// host entry point (csrc/k_tile.hip)
int launch_tile(bool full, bool two, const TileArgs& a, int grid, int threads, size_t smem, hipStream_t stream, bool w32 = false);
// resident workgroups per CU of that instantiation (LDS, registers, wave slots)
int tile_occupancy(bool full, bool two, int nch, int threads, size_t smem, bool w32 = false);

}  // namespace wfa
