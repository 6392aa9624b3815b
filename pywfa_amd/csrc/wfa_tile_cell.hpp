// wfa_tile_cell.hpp — the arithmetic of the tiled wide-wavefront kernel (wfa_tile.hpp) that does not depend on how a wave
// runs it: ring slots, the super-step's diagonal ranges, what a tile loads and writes back, and compute-next of one cell.
// Plain functions, compiled for the device by hipcc and for the host by g++ (tools/tile_model.cpp: the host model of the
// blocked schedule that tests/test_tile_model.py checks against the CPU reference restatement — the exactness argument of the schedule, run on CPU).
//
// R = /root/reference/pywfa/WFA2_lib/wavefront.  Scores advance in steps of g = gcd of the penalties; X, OE, E (, OE2, E2)
// are the penalties in steps.  Step t holds the wavefronts of score t g (R/wavefront_compute_affine.c:44-86,
// R/wavefront_compute_affine2p.c:45-106).
//
// The blocked schedule.  The diagonals of a pair are cut into fixed blocks of Bw columns (column c = k + plen).  A super-step
// advances every active block by T score steps: a wave loads the tile of its block — the block's columns plus a halo of T
// columns per side, for every row the T steps will read — into LDS, runs the T steps there (the dependency cone of a cell
// widens by at most one diagonal per step, so after T steps exactly the block's own columns are still exact), and writes the
// rows later super-steps need back to the HBM workspace: rows written once and read once per T steps instead of every step.
//
// Why that is exact although the reference trims every wavefront to its first / last in-bounds cell after every step
// (R/wavefront_compute.c:571-605), which looks like a row-wide dependency:
//  * a cell outside the trimmed limits of its row is dead (negative: NULL plus the steps since) or lies past the end of a
//    sequence.  Dead cells behave like NULL in every max(), so computing a superset of the reference's range changes nothing;
//  * a gap cell PAST THE END outside its row's trimmed limits is the one case where trimming changes a value (the reference
//    sets it to NULL, untrimmed it would propagate).  The kernel notes cells whose untrimmed M candidate passes the end
//    ("taint"; a superset of the gap cells past the end), repeats such a super-step collecting, per gap row, the first / last
//    in-bounds and past-the-end diagonals, and hands the pair to the step-by-step kernel (wfa_wide.hpp) when a past-the-end
//    cell lies outside [first, last in-bounds] of its row.  In long reads these cells appear only around the end diagonal in the
//    last steps, between thousands of in-bounds cells;
//  * diagonals outside [-plen, tlen] hold no cell: always trimmed, so they are NULL here too.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WFA_TILE_HD __host__ __device__ __forceinline__
#else
#define WFA_TILE_HD static inline
#endif

// NULL of the int16 rows.  Round 6: -32768 (was -16384): offsets are loaded into ints before any arithmetic, so nothing needs the headroom
// of a half-range NULL, and with the whole positive range for offsets the tiled kernel takes reads of up to 32 000 bases (round 5:
// 16 000; exact 30 kb ran the step-by-step wide kernel).  A dead value gains at most 1 per step and must stay negative: 32 000 steps.
#define WFA_TILE_NULL (-32768)
#define WFA_TILE_NULL2 0x80008000u   // two of them
#define WFA_TILE_MAX_LEN 32000       // longest sequence, and most steps, of the int16 rows

namespace wfa {

struct TileGeom {
  int X, OE, E, OE2, E2;   // penalties in steps (OE2 = 0: gap-affine)
  int T;                   // steps per super-step (even)
  int Wt;                  // tile width in columns (multiple of 64), Bw = Wt - 2 T columns are the block's own
  int DM;                  // max(X, OE, OE2): the oldest M row a step reads
};

WFA_TILE_HD int tile_max(int a, int b) { return a > b ? a : b; }
WFA_TILE_HD int tile_min(int a, int b) { return a < b ? a : b; }

WFA_TILE_HD int tile_bw(const TileGeom& g) { return g.Wt - 2 * g.T; }
// "Far" form of the M rows in LDS (gap-affine-2p with a deep second gap opening, e.g. 4/6/2/24/1: lags 4, 8 and 25): the ring in LDS
// only spans the near lags X and OE, and the T rows read at lag OE2 — all of them older than the super-step — sit in a linear
// buffer, row j for step t0 + j.  27 instead of 36 LDS rows for pywfa's defaults at T = 8.  Conditions: OE2 is the one far lag,
// none of its rows is also a near row, and the ring is deep enough for the T new rows a super-step writes back.
WFA_TILE_HD bool tile_far(const TileGeom& g) {
  const int near = tile_max(g.X, g.OE);
  return g.OE2 > 0 && g.OE2 - g.T >= near && g.T <= near + 1;
}
WFA_TILE_HD int tile_m_ring(const TileGeom& g) { return tile_far(g) ? tile_max(g.X, g.OE) + 1 : g.DM + 1; }
// LDS rows of a tile: the M ring, I1 / D1 rings of E + 1, I2 / D2 rings of E2 + 1, then (far form) T rows of M at lag OE2
WFA_TILE_HD int tile_lds_rows(const TileGeom& g) {
  return tile_m_ring(g) + 2 * (g.E + 1) + (g.OE2 > 0 ? 2 * (g.E2 + 1) : 0) + (tile_far(g) ? g.T : 0);
}
WFA_TILE_HD int tile_lds_pitch(const TileGeom& g) { return g.Wt + 4; }   // halfs: two guard columns per side (dword-aligned column 0)
// first LDS row of a component's ring (0 M, 1 I1, 2 D1, 3 I2, 4 D2) and its depth
WFA_TILE_HD int tile_lds_ring_base(const TileGeom& g, int comp) {
  const int nm = tile_m_ring(g), n1 = g.E + 1, n2 = g.E2 + 1;
  return comp == 0 ? 0 : comp == 1 ? nm : comp == 2 ? nm + n1 : comp == 3 ? nm + 2 * n1 : nm + 2 * n1 + n2;
}
WFA_TILE_HD int tile_lds_far_base(const TileGeom& g) { return tile_m_ring(g) + 2 * (g.E + 1) + 2 * (g.E2 + 1); }
WFA_TILE_HD int tile_lds_ring_depth(const TileGeom& g, int comp) { return comp == 0 ? tile_m_ring(g) : comp <= 2 ? g.E + 1 : g.E2 + 1; }
WFA_TILE_HD int tile_lds_slot(const TileGeom& g, int comp, int t) {   // t >= -4 depth
  const int d = tile_lds_ring_depth(g, comp);
  return tile_lds_ring_base(g, comp) + (t + 8 * d) % d;
}
// LDS row that holds M[t0 - d] for a super-step starting at t0 (d = 1 .. DM): the ring, or (far form, d beyond the near lags)
// the far buffer's row j = OE2 - d, read at step t0 + j
WFA_TILE_HD int tile_lds_slot_m_old(const TileGeom& g, int t0, int d) {
  if (tile_far(g) && d > tile_max(g.X, g.OE)) return tile_lds_far_base(g) + (g.OE2 - d);
  return tile_lds_slot(g, 0, t0 - d);
}
// LDS row a step reads M at lag OE2 from: step j of the super-step starting at t0
WFA_TILE_HD int tile_lds_slot_m_far_in(const TileGeom& g, int t0, int j) {
  return tile_far(g) ? tile_lds_far_base(g) + j : tile_lds_slot(g, 0, t0 + j - g.OE2);
}

// HBM rows of a component: the D = lag rows a super-step starts from and the rows it leaves for the next one must not share
// slots (other blocks of the same super-step still read the old rows): D + T consecutive rows modulo D + T when T < D, two
// halves of D rows alternating by super-step otherwise.  2 D rows are allocated either way.
WFA_TILE_HD int tile_hbm_lag(const TileGeom& g, int comp) { return comp == 0 ? g.DM : comp <= 2 ? g.E : g.E2; }
WFA_TILE_HD int tile_hbm_ring_base(const TileGeom& g, int comp) {
  return comp == 0 ? 0 : comp == 1 ? 2 * g.DM : comp == 2 ? 2 * g.DM + 2 * g.E : comp == 3 ? 2 * g.DM + 4 * g.E : 2 * g.DM + 4 * g.E + 2 * g.E2;
}
WFA_TILE_HD int tile_hbm_rows(const TileGeom& g) { return 2 * g.DM + 4 * g.E + (g.OE2 > 0 ? 4 * g.E2 : 0); }
WFA_TILE_HD int tile_hbm_slot(const TileGeom& g, int comp, int t) {   // t >= 0
  const int D = tile_hbm_lag(g, comp);
  const int s = (g.T >= D) ? (((t / g.T) & 1) * D + t % D) : (t % (D + g.T));
  return tile_hbm_ring_base(g, comp) + s;
}
// does a super-step starting at t0 read row t0 - d of this component from HBM (d = 1 .. lag)?
WFA_TILE_HD bool tile_loads_row(const TileGeom& g, int comp, int d) {
  if (comp != 0) return true;   // (gap rings: every row of the last E / E2 steps is read)
  // M[t0 - d] is read at step t0 + j with lag L iff d = L - j for some 0 <= j < T
  if (g.X > 0 && d <= g.X && d > g.X - g.T) return true;
  if (g.OE > 0 && d <= g.OE && d > g.OE - g.T) return true;
  if (g.OE2 > 0 && d <= g.OE2 && d > g.OE2 - g.T) return true;
  return false;
}
// does the super-step write row t0 + j back (j = 0 .. T-1)?  The last `lag` rows: what later super-steps can still read.
WFA_TILE_HD bool tile_keeps_row(const TileGeom& g, int comp, int j) { return j >= g.T - tile_hbm_lag(g, comp); }

// The largest diagonal distance from the start a score of t steps can pay for: one gap of the cheaper kind
// (R/wavefront_compute.c:40-86 applied to untrimmed limits gives exactly this hull).
WFA_TILE_HD int tile_reach(const TileGeom& g, int t) {
  const int r1 = (t >= g.OE) ? 1 + (t - g.OE) / g.E : 0;
  const int r2 = (g.OE2 > 0 && t >= g.OE2) ? 1 + (t - g.OE2) / g.E2 : 0;
  return tile_max(r1, r2);
}
WFA_TILE_HD int tile_lo(const TileGeom& g, int t, int plen, int pbf) { return tile_max(-plen, -pbf - tile_reach(g, t)); }
WFA_TILE_HD int tile_hi(const TileGeom& g, int t, int tlen, int tbf) { return tile_min(tlen, tbf + tile_reach(g, t)); }

// compute-next of one cell.  Inputs are row values (NULL / dead values are negative).  Out: the five components (M not yet
// clamped: m_raw), and the piggy-back origin code the backtrace would choose (R/wavefront_backtrace.c:49-59: on equal offsets
// mismatch > D2 > D1 > I2 > I1, extension > opening; same encoding as wfa_wide.hpp / wfa_general.hpp PB).
struct TileCell { int m_raw, i1, d1, i2, d2, code; };

template <bool TWO, bool FULL>
WFA_TILE_HD TileCell tile_cell(int mx, int mo_lo, int mo_hi, int ie_lo, int de_hi, int mo2_lo, int mo2_hi, int i2e_lo, int d2e_hi) {
  TileCell c;
  c.i1 = tile_max(mo_lo, ie_lo) + 1;
  c.d1 = tile_max(mo_hi, de_hi);
  const int x1 = mx + 1;
  c.i2 = WFA_TILE_NULL; c.d2 = WFA_TILE_NULL; c.code = 0;
  if (TWO) {
    c.i2 = tile_max(mo2_lo, i2e_lo) + 1;
    c.d2 = tile_max(mo2_hi, d2e_hi);
    const int best = tile_max(tile_max(c.d1, c.d2), tile_max(x1, tile_max(c.i1, c.i2)));
    c.m_raw = best;
    if (FULL) {
      const int mc = (x1 >= best) ? 0 : (c.d2 >= best) ? 2 : (c.d1 >= best) ? 1 : (c.i2 >= best) ? 4 : 3;
      c.code = mc | ((ie_lo >= mo_lo) ? 8 : 0) | ((de_hi >= mo_hi) ? 16 : 0) | ((i2e_lo >= mo2_lo) ? 32 : 0) | ((d2e_hi >= mo2_hi) ? 64 : 0);
    }
  } else {
    c.m_raw = tile_max(c.d1, tile_max(x1, c.i1));
    if (FULL) {
      const int mc = (x1 >= tile_max(c.d1, c.i1)) ? 0 : ((c.d1 >= c.i1) ? 1 : 2);
      c.code = mc | ((ie_lo >= mo_lo) ? 4 : 0) | ((de_hi >= mo_hi) ? 8 : 0);
    }
  }
  return c;
}

}  // namespace wfa
