// tile_model.cpp — host model of the blocked ("tiled") schedule of csrc/wfa_tile.hpp, one pair at a time, sequentially.
// Test infrastructure: it shares csrc/wfa_tile_cell.hpp (ring slots, ranges, what a tile loads / keeps, compute-next of a cell)
// with the kernel and restates the rest of the schedule — tile load with halo, T steps in the tile, write-back, the taint /
// trimming check, the termination key, piggy-back codes and their walk — so that tests/test_tile_model.py can check on the CPU,
// against the oracle, that the schedule computes the reference's scores and op strings (or hands the pair on), for every
// geometry (T, tile width) and penalty set.  LDS slots the schedule claims it never reads are poisoned.
//
//   g++ -O2 -shared -fPIC -I pywfa_amd/csrc tools/tile_model.cpp -o /tmp/libtile_model.so
#include <limits.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "wfa_tile_cell.hpp"

using namespace wfa;

namespace {

const int16_t POISON = 12345;

struct Model {
  TileGeom g;
  bool two, full;
  int ef, pbf, pef, tbf, tef;
  int max_steps;   // in steps (INT_MAX: none)
  const uint8_t *P, *T_;
  int plen, tlen;
};

}  // namespace

// status: 0 reached (score steps in *end_t, op string in ops / *nops), 3 handed on, 4 step limit
extern "C" int tile_model_align(int X, int OE, int E, int OE2, int E2, int T, int Wt, int full, int ef, int pbf, int pef, int tbf, int tef,
                                int max_steps, const uint8_t* P, int plen, const uint8_t* Tx, int tlen, int* end_t_out, int* end_k_out,
                                uint8_t* ops, int* nops, int* careful_passes, int force_careful) {
  Model m;
  m.g.X = X; m.g.OE = OE; m.g.E = E; m.g.OE2 = OE2; m.g.E2 = E2; m.g.T = T; m.g.Wt = Wt;
  m.g.DM = tile_max(tile_max(X, OE), OE2);
  m.two = OE2 > 0; m.full = full != 0;
  const TileGeom& g = m.g;
  const int Bw = tile_bw(g);
  const int NC = m.two ? 5 : 3;
  if (!ef) { pbf = pef = tbf = tef = 0; }
  const int ak = tlen - plen;
  const int ncols = plen + tlen + 1;
  const int nb = (ncols + Bw - 1) / Bw;
  const int rwh = nb * Bw + 2 * T;                       // elements per HBM row: element = column + T
  std::vector<std::vector<int16_t>> hbm(tile_hbm_rows(g), std::vector<int16_t>(rwh, (int16_t)WFA_TILE_NULL));
  const int pitch = tile_lds_pitch(g);
  const int nlds = tile_lds_rows(g);
  std::vector<int16_t> lds((size_t)nlds * pitch);
  std::vector<uint8_t> codes;
  std::vector<int> dir;   // lo, hi, base per step
  *careful_passes = 0;
  int end_reason = 0, end_t = 0, end_k = 0;
  if (plen > WFA_TILE_MAX_LEN || tlen > WFA_TILE_MAX_LEN) return 3;

  for (int ss = 0; !end_reason; ++ss) {
    const int t0 = ss * T;
    if (t0 + T > WFA_TILE_MAX_LEN) { end_reason = 3; break; }
    // directory of the T steps
    int lo_j[64], hi_j[64]; long long base_j[64];
    for (int j = 0; j < T; ++j) {
      lo_j[j] = tile_lo(g, t0 + j, plen, pbf); hi_j[j] = tile_hi(g, t0 + j, tlen, tbf);
      base_j[j] = (long long)codes.size();
      if (m.full) { codes.resize(codes.size() + (size_t)(hi_j[j] - lo_j[j] + 1), 0xEE); dir.push_back(lo_j[j]); dir.push_back(hi_j[j]); dir.push_back((int)base_j[j]); }
    }
    const int clo = lo_j[T - 1] + plen, chi = hi_j[T - 1] + plen;
    const int b_first = clo / Bw, b_last = chi / Bw;
    unsigned endkey = 0xFFFFFFFFu;
    bool taint = false;
    for (int pass = 0; pass < 2; ++pass) {
      const bool careful = pass == 1 || force_careful;
      // per (step, gap component): first / last in-bounds and past-the-end diagonal
      std::vector<int> in_min(T * 4, INT_MAX), in_max(T * 4, INT_MIN), pe_min(T * 4, INT_MAX), pe_max(T * 4, INT_MIN);
      // writes of a super-step go to a copy: blocks of one super-step never read each other's new rows (the ring-slot rule);
      // the model checks that rule by reading from the state before the super-step only
      std::vector<std::vector<int16_t>> hbm_new = hbm;
      for (int b = b_first; b <= b_last; ++b) {
        for (size_t i = 0; i < lds.size(); ++i) lds[i] = POISON;
        // guards
        for (int r = 0; r < nlds; ++r) { int16_t* row = &lds[(size_t)r * pitch]; row[0] = row[1] = row[pitch - 2] = row[pitch - 1] = (int16_t)WFA_TILE_NULL; }
        // tile load
        for (int comp = 0; comp < NC; ++comp) {
          const int lag = tile_hbm_lag(g, comp);
          for (int d = 1; d <= lag; ++d) {
            if (!tile_loads_row(g, comp, d)) continue;
            int16_t* row = &lds[(size_t)(comp == 0 ? tile_lds_slot_m_old(g, t0, d) : tile_lds_slot(g, comp, t0 - d)) * pitch + 2];
            if (t0 - d < 0) { for (int c = 0; c < Wt; ++c) row[c] = (int16_t)WFA_TILE_NULL; continue; }
            const int16_t* src = &hbm[tile_hbm_slot(g, comp, t0 - d)][(size_t)b * Bw];
            for (int c = 0; c < Wt; ++c) row[c] = src[c];
          }
        }
        for (int j = 0; j < T; ++j) {
          const int t = t0 + j;
          auto rowp = [&](int comp, int tt) { return &lds[(size_t)tile_lds_slot(g, comp, tt) * pitch + 2]; };
          const int16_t* pX = rowp(0, t - g.X); const int16_t* pO = rowp(0, t - g.OE);
          const int16_t* pI = rowp(1, t - g.E); const int16_t* pD = rowp(2, t - g.E);
          const int16_t* pO2 = m.two ? &lds[(size_t)tile_lds_slot_m_far_in(g, t0, j) * pitch + 2] : nullptr;
          const int16_t* pI2 = m.two ? rowp(3, t - g.E2) : nullptr; const int16_t* pD2 = m.two ? rowp(4, t - g.E2) : nullptr;
          std::vector<int16_t> out[5];
          for (int c = 0; c < NC; ++c) out[c].assign(Wt, 0);
          for (int col = 0; col < Wt; ++col) {
            const int c = b * Bw - T + col, k = c - plen;
            const bool owned = col >= T && col < T + Bw;
            // a cell inside the step's exact region must never read a poisoned value
            const bool exact = col >= j + 1 - 1 && col <= Wt - 1 - j;   // (columns j .. Wt-1-j are exact at step j)
            int v[9];
            v[0] = pX[col]; v[1] = pO[col - 1]; v[2] = pO[col + 1]; v[3] = pI[col - 1]; v[4] = pD[col + 1];
            v[5] = m.two ? pO2[col - 1] : WFA_TILE_NULL; v[6] = m.two ? pO2[col + 1] : WFA_TILE_NULL;
            v[7] = m.two ? pI2[col - 1] : WFA_TILE_NULL; v[8] = m.two ? pD2[col + 1] : WFA_TILE_NULL;
            if (exact) for (int i = 0; i < 9; ++i) if (v[i] == POISON) return -1000 - i;   // the schedule read a row it did not load
            TileCell cc = m.two ? (m.full ? tile_cell<true, true>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]) : tile_cell<true, false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]))
                                : (m.full ? tile_cell<false, true>(v[0], v[1], v[2], v[3], v[4], 0, 0, 0, 0) : tile_cell<false, false>(v[0], v[1], v[2], v[3], v[4], 0, 0, 0, 0));
            if (t == 0 && k >= -pbf && k <= tbf) cc.m_raw = tile_max(k, 0);   // R/wavefront_aligner.c:251-310
            int v5[5] = {cc.m_raw, cc.i1, cc.d1, cc.i2, cc.d2};
            const int limk = tile_min(tlen, plen + k), base = tile_max(k, 0);
            const bool kin = limk >= base && k >= -plen && k <= tlen;
            if (!kin) for (int q = 0; q < 5; ++q) v5[q] = WFA_TILE_NULL;   // a diagonal without cells: always trimmed
            const bool m_in = kin && v5[0] >= base && v5[0] <= limk;
            if (owned && kin && cc.m_raw > limk) taint = true;
            if (careful && owned && kin) {
              for (int q = 1; q < NC; ++q) {
                const int idx = j * 4 + (q - 1);
                if (v5[q] >= base && v5[q] <= limk) { in_min[idx] = tile_min(in_min[idx], k); in_max[idx] = tile_max(in_max[idx], k); }
                else if (v5[q] > limk) { pe_min[idx] = tile_min(pe_min[idx], k); pe_max[idx] = tile_max(pe_max[idx], k); }
              }
            }
            if (!m_in) v5[0] = WFA_TILE_NULL;
            else {
              int h = v5[0], vv = h - k;
              while (vv < plen && h < tlen && P[vv] == Tx[h]) { ++vv; ++h; }
              v5[0] = h;
              if (owned) {
                bool fin = ef ? ((h >= tlen && plen - vv <= pef) || (vv >= plen && tlen - h <= tef)) : (k == ak && h >= tlen);
                if (fin) { const unsigned key = ((unsigned)t << 16) | (unsigned)(k + 32768); if (key < endkey) endkey = key; }
              }
            }
            for (int q = 0; q < NC; ++q) out[q][col] = (int16_t)v5[q];
            if (m.full && owned && k >= lo_j[j] && k <= hi_j[j]) codes[(size_t)base_j[j] + (k - lo_j[j])] = (uint8_t)cc.code;
          }
          for (int q = 0; q < NC; ++q) { int16_t* w = rowp(q, t); for (int col = 0; col < Wt; ++col) w[col] = out[q][col]; }
        }
        // write-back of the block's own columns
        for (int comp = 0; comp < NC; ++comp)
          for (int j = 0; j < T; ++j) {
            if (!tile_keeps_row(g, comp, j)) continue;
            const int16_t* row = &lds[(size_t)tile_lds_slot(g, comp, t0 + j) * pitch + 2];
            int16_t* dst = &hbm_new[tile_hbm_slot(g, comp, t0 + j)][(size_t)b * Bw];
            for (int col = T; col < T + Bw; ++col) dst[col] = row[col];
          }
      }
      // the ring-slot rule: no row a block of this super-step reads was rewritten by another block
      for (int comp = 0; comp < NC; ++comp)
        for (int d = 1; d <= tile_hbm_lag(g, comp); ++d)
          if (t0 - d >= 0 && tile_loads_row(g, comp, d) && hbm_new[tile_hbm_slot(g, comp, t0 - d)] != hbm[tile_hbm_slot(g, comp, t0 - d)]) return -2000;
      if (!careful && taint) { ++*careful_passes; continue; }   // repeat with the trimming statistics
      if (careful) {
        // (steps after the one that reached the end do not count: nothing the reference never computed can be wrong)
        const int j_end = (endkey != 0xFFFFFFFFu) ? (int)(endkey >> 16) - t0 : T - 1;
#ifndef TILE_MODEL_NO_CHECK
        for (int i = 0; i < T * 4; ++i)
          if (i / 4 <= j_end && pe_min[i] != INT_MAX && (in_min[i] == INT_MAX || pe_min[i] < in_min[i] || pe_max[i] > in_max[i])) end_reason = 3;
#endif
      }
      hbm.swap(hbm_new);
      break;
    }
    if (end_reason) break;
    // step limit (R/wavefront_unialign.c:98-107: tested after compute-next of a score, before its extension) vs the end
    const int t_end = (endkey != 0xFFFFFFFFu) ? (int)(endkey >> 16) : INT_MAX;
    int t_lim = INT_MAX;
    if (max_steps != INT_MAX) { t_lim = tile_max(1, max_steps); }
    if (t_lim < t0 + T && t_lim <= t_end) { end_reason = 4; break; }
    if (t_end != INT_MAX) { end_reason = 1; end_t = t_end; end_k = (int)(endkey & 0xFFFFu) - 32768; break; }
  }
  *end_t_out = end_t; *end_k_out = end_k; *nops = 0;
  if (end_reason != 1 || !m.full) return end_reason;
  // walk the codes back (as wfa_wide.hpp), unpack forwards
  std::vector<uint8_t> ev;
  int tc = end_t, k = end_k, comp = 0;
  while (tc > 0) {
    const int* d = &dir[(size_t)tc * 3];
    const int cd = (k >= d[0] && k <= d[1]) ? codes[(size_t)d[2] + (k - d[0])] : 0;
    const uint8_t flag = (comp == 0) ? 0x80 : 0;
    int src;
    if (m.two) src = (comp == 0) ? (cd & 7) : (comp == 1) ? 3 : (comp == 2) ? 1 : (comp == 3) ? 4 : 2;
    else src = (comp == 0) ? ((cd & 3) == 0 ? 0 : ((cd & 3) == 1 ? 1 : 3)) : (comp == 1 ? 3 : 1);
    const int bi1 = m.two ? 8 : 4, bd1 = m.two ? 16 : 8;
    if (src == 0) { ev.push_back((uint8_t)('X' | 0x80)); tc -= g.X; }
    else if (src == 1) { ev.push_back((uint8_t)('D' | flag)); ++k; if (cd & bd1) { tc -= g.E; comp = 2; } else { tc -= g.OE; comp = 0; } }
    else if (src == 2) { ev.push_back((uint8_t)('D' | flag)); ++k; if (cd & 64) { tc -= g.E2; comp = 4; } else { tc -= g.OE2; comp = 0; } }
    else if (src == 3) { ev.push_back((uint8_t)('I' | flag)); --k; if (cd & bi1) { tc -= g.E; comp = 1; } else { tc -= g.OE; comp = 0; } }
    else { ev.push_back((uint8_t)('I' | flag)); --k; if (cd & 32) { tc -= g.E2; comp = 3; } else { tc -= g.OE2; comp = 0; } }
    if (ev.size() > (size_t)(plen + tlen + 8)) return -3000;
  }
  if (tc < 0) return -3001;
  int n = 0;
  auto emit = [&](char c, int cnt) { for (int i = 0; i < cnt; ++i) ops[n++] = (uint8_t)c; };
  auto lcp = [&](int v, int h) { int r = 0; while (v + r < plen && h + r < tlen && P[v + r] == Tx[h + r]) ++r; return r; };
  int h = tile_max(k, 0), v = h - k;
  emit('I', h); emit('D', v);
  { const int e = lcp(v, h); emit('M', e); v += e; h += e; }
  for (long long e_ = (long long)ev.size() - 1; e_ >= 0; --e_) {
    const int op = ev[e_] & 0x7F;
    if (op == 'X') { emit('X', 1); ++v; ++h; }
    else if (op == 'I') { emit('I', 1); ++h; }
    else { emit('D', 1); ++v; }
    if (ev[e_] & 0x80) { const int e = lcp(v, h); emit('M', e); v += e; h += e; }
  }
  emit('I', tlen - h); emit('D', plen - v);
  *nops = n;
  return 1;
}
