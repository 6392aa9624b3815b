/*
 * oracle/wfa_oracle.c — TEST INFRASTRUCTURE (the parity checker), not product code.
 *
 * A plain-C, single-threaded CPU restatement of the algorithm on pywfa's hot path: the
 * gap-affine / gap-affine-2p wavefront alignment loop of the vendored WFA2-lib v2.3
 * (R = /root/reference/pywfa/WFA2_lib/wavefront).  It is written from the algorithm's
 * semantics, with flat arrays and an explicit [lo,hi]+NULL model,
 * not from WFA2-lib's slab/component structure.  Each function cites the reference
 * file:line whose behaviour it restates.
 *
 * PINNED: oracle/_ref (the real WFA2-lib compiled from /root/reference by oracle/Makefile)
 * is compared with this file bit-for-bit (status, score, op string) on large random
 * corpora by tests/test_oracle_vs_ref.py, and both are compared with the committed golden
 * fixtures in tests/golden/ (the reference's own known answers, tests/test.py) by
 * tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 * Nothing under pywfa_amd/ links or imports it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#include "wfa_hip.h"

#define OFFSET_NULL (INT32_MIN / 2) /* R/wavefront_offset.h:44 */

/* internal end reasons (R/wfa.h:52-55) */
#define END_REACHED 1
#define END_UNREACHABLE 2

/* one stored wavefront: offsets for k in [base, base+cap), valid range [lo,hi] */
typedef struct {
  int exists; /* pointer != NULL in the reference */
  int lo, hi; /* lo > hi  <=>  ->null flag of the reference */
  int base;   /* k of element 0 */
  int64_t idx; /* index of element 0 in the arena */
} wf_t;

typedef struct {
  /* penalties after wavefront_penalties_set_* (R/wavefront_penalties.c:95-173) */
  int match, x, o1, e1, o2, e2;
  int ncomp; /* 1 indel/edit/linear, 3 affine, 5 affine2p */
  int metric; /* WFA_DIST_* */
  int scope; /* max_score_scope (R/wavefront_components.c:81-124) */
  /* match < 0 with free begins: the begin cells are (re)seeded score by score (R/wavefront_compute.c:124-254) */
  int ef_seed, ef_pbf, ef_tbf;
  /* history: comp-major tables indexed by score (full) or score % scope (score-only) */
  wf_t* wf[5]; /* 0=M 1=I1 2=D1 3=I2 4=D2 */
  int64_t wf_cap;
  int modular;
  int32_t* arena;
  int64_t arena_cap, arena_used;
  int64_t slot_stride; /* modular: ints per (slot, comp) */
  /* scratch for heuristics */
  int32_t* tmp;
  int64_t tmp_cap;
} oracle_ws_t;

static void ws_free(oracle_ws_t* ws) {
  int c;
  for (c = 0; c < 5; ++c) free(ws->wf[c]);
  free(ws->arena);
  free(ws->tmp);
  memset(ws, 0, sizeof(*ws));
}

static int ws_reserve_scores(oracle_ws_t* ws, int64_t n) {
  if (n <= ws->wf_cap) return 0;
  int64_t cap = ws->wf_cap ? ws->wf_cap : 64;
  while (cap < n) cap *= 2;
  int c;
  for (c = 0; c < 5; ++c) {
    wf_t* p = (wf_t*)realloc(ws->wf[c], (size_t)cap * sizeof(wf_t));
    if (!p) return -1;
    memset(p + ws->wf_cap, 0, (size_t)(cap - ws->wf_cap) * sizeof(wf_t));
    ws->wf[c] = p;
  }
  ws->wf_cap = cap;
  return 0;
}

static int ws_reserve_arena(oracle_ws_t* ws, int64_t n) {
  if (n <= ws->arena_cap) return 0;
  int64_t cap = ws->arena_cap ? ws->arena_cap : 4096;
  while (cap < n) cap *= 2;
  int32_t* p = (int32_t*)realloc(ws->arena, (size_t)cap * sizeof(int32_t));
  if (!p) return -1;
  ws->arena = p;
  ws->arena_cap = cap;
  return 0;
}

/* storage for wavefront (comp c, score s) covering k in [alo,ahi], valid range [lo,hi] */
static int wf_alloc_range(oracle_ws_t* ws, int c, int s, int alo, int ahi, int lo, int hi);
static int wf_alloc(oracle_ws_t* ws, int c, int s, int lo, int hi) { return wf_alloc_range(ws, c, s, lo, hi, lo, hi); }
static int wf_alloc_range(oracle_ws_t* ws, int c, int s, int alo, int ahi, int lo, int hi) {
  const int64_t slot = ws->modular ? (s % ws->scope) : s;
  wf_t* w = &ws->wf[c][slot];
  const int64_t n = (int64_t)ahi - alo + 1;
  if (ws->modular) {
    w->idx = (slot * 5 + c) * ws->slot_stride;
    if (n > ws->slot_stride) return -1;
  } else {
    if (ws_reserve_arena(ws, ws->arena_used + n)) return -1;
    w->idx = ws->arena_used;
    ws->arena_used += n;
  }
  w->exists = 1;
  w->lo = lo;
  w->hi = hi;
  w->base = alo;
  return 0;
}

static inline wf_t* wf_slot(oracle_ws_t* ws, int c, int s) {
  return &ws->wf[c][ws->modular ? (s % ws->scope) : s];
}

/* R/wavefront_compute.c:255-297: a wavefront at a negative score, missing, or flagged
 * null is replaced by the shared null wavefront (lo=1, hi=-1; R/wavefront.c:110-117). */
typedef struct {
  int null;
  int lo, hi;
  const int32_t* off; /* off[k - base] */
  int base;
} wf_in_t;

static wf_in_t wf_fetch(oracle_ws_t* ws, int c, int s) {
  wf_in_t in;
  in.null = 1;
  in.lo = 1;
  in.hi = -1;
  in.off = NULL;
  in.base = 0;
  if (s < 0) return in;
  const wf_t* w = wf_slot(ws, c, s);
  if (!w->exists || w->lo > w->hi) return in;
  in.null = 0;
  in.lo = w->lo;
  in.hi = w->hi;
  in.off = ws->arena + w->idx;
  in.base = w->base;
  return in;
}

/* Reading outside [lo,hi] yields NULL (lazy padding, R/wavefront_compute.c:490-567). */
static inline int32_t wf_get(const wf_in_t* in, int k) {
  return (k >= in->lo && k <= in->hi) ? in->off[k - in->base] : OFFSET_NULL;
}

/* Work counters for bench.py's "offsets/s" figure: offsets of M computed, offsets of all components
 * computed, bases compared by the extension, since the last wfa_oracle_counters(NULL) call.  Not thread-safe. */
static int64_t g_m_offsets, g_all_offsets, g_bases;
void wfa_oracle_counters(int64_t* out3) {
  if (out3) { out3[0] = g_m_offsets; out3[1] = g_all_offsets; out3[2] = g_bases; }
  else { g_m_offsets = 0; g_all_offsets = 0; g_bases = 0; }
}

#define MAX2(a, b) ((a) > (b) ? (a) : (b))
#define MIN2(a, b) ((a) < (b) ? (a) : (b))

/* R/wavefront_compute.c:571-605 (wavefront_compute_trim_ends) */
static void wf_trim(oracle_ws_t* ws, wf_t* w, int plen, int tlen) {
  const int32_t* off = ws->arena + w->idx;
  int k;
  for (k = w->hi; k >= w->lo; --k) {
    const int32_t o = off[k - w->base];
    const uint32_t h = (uint32_t)o;
    const uint32_t v = (uint32_t)(o - k);
    if (h <= (uint32_t)tlen && v <= (uint32_t)plen) break;
  }
  w->hi = k;
  for (k = w->lo; k <= w->hi; ++k) {
    const int32_t o = off[k - w->base];
    const uint32_t h = (uint32_t)o;
    const uint32_t v = (uint32_t)(o - k);
    if (h <= (uint32_t)tlen && v <= (uint32_t)plen) break;
  }
  w->lo = k;
}

/*
 * match < 0 with free begins (R/wavefront_compute.c:124-254).  With a match score the begin-free cells are not all worth the
 * same, so wavefront 0 holds diagonal 0 only and the cell that skips j text (pattern) bases enters the M wavefront of score
 * j * (-match): (k = j, offset j) resp. (k = -j, offset 0), if it beats what compute-next put there.
 *   required   :124-139     limits :140-153     re-seeding a computed wavefront :171-213     a null step :214-254
 * One branch of the reference is undefined: a null step at a required score whose j exceeds both free begins gets lo = hi = 0
 * with offsets[0] never written (:229-251).  Here that offset is NULL (the wavefront is null) and the event is counted
 * (wfa_oracle_undefined_reads): parity against the real library is claimed only where the count stays 0.
 */
static int64_t g_undefined_reads;
int64_t wfa_oracle_undefined_reads(int reset) { const int64_t v = g_undefined_reads; if (reset) g_undefined_reads = 0; return v; }

static inline int ef_required(const oracle_ws_t* ws, int s) {
  return ws->ef_seed && (s % (-ws->match)) == 0;
}
/* a null step at a required score (wavefront_compute_endsfree_allocate_null); returns -1 when out of memory */
static int ef_seed_null_step(oracle_ws_t* ws, int s) {
  const int ek = s / (-ws->match);
  const int tb = ws->ef_tbf >= ek, pb = ws->ef_pbf >= ek;
  int lo = 0, hi = 0;
  if (tb && pb) { lo = -ek; hi = ek; } else if (tb) { lo = hi = ek; } else if (pb) { lo = hi = -ek; }
  if (!tb && !pb) { ++g_undefined_reads; return 0; }   /* the reference: lo = hi = 0, offsets[0] unset; here: null */
  if (wf_alloc(ws, 0, s, lo, hi)) return -1;
  wf_t* w = wf_slot(ws, 0, s);
  int32_t* o = ws->arena + w->idx;
  int k;
  for (k = lo; k <= hi; ++k) o[k - w->base] = OFFSET_NULL;
  if (tb) o[ek - w->base] = ek;
  if (pb) o[-ek - w->base] = 0;
  return 0;
}
/* storage range of a computed wavefront at a required score (wavefront_compute_endsfree_limits) */
static void ef_limits(const oracle_ws_t* ws, int s, int* alo, int* ahi) {
  const int ek = s / (-ws->match);
  if (ws->ef_tbf >= ek && ek > *ahi) *ahi = ek;
  if (ws->ef_pbf >= ek && -ek < *alo) *alo = -ek;
}
/* re-seeding of the computed M wavefront before its ends are trimmed (wavefront_compute_endsfree_init) */
static void ef_seed_computed(oracle_ws_t* ws, wf_t* w, int s) {
  const int ek = s / (-ws->match);
  const int lo = w->lo, hi = w->hi;
  int32_t* o = ws->arena + w->idx;
  int k;
  if (ws->ef_tbf >= ek) {
    if (hi >= ek) {
      /* (ek < lo: the reference compares with a cell outside the wavefront and may write it; it stays outside, unobservable) */
      if (ek >= lo && o[ek - w->base] <= ek) o[ek - w->base] = ek;
    } else {
      for (k = hi + 1; k < ek; ++k) o[k - w->base] = OFFSET_NULL;
      o[ek - w->base] = ek;
      w->hi = ek;
    }
  }
  if (ws->ef_pbf >= ek) {
    if (lo <= -ek) {
      if (-ek <= hi && o[-ek - w->base] <= 0) o[-ek - w->base] = 0;
    } else {
      o[-ek - w->base] = 0;
      for (k = -ek + 1; k < lo; ++k) o[k - w->base] = OFFSET_NULL;
      w->lo = -ek;
    }
  }
}

/*
 * Compute-next for score s (R/wavefront_compute_affine.c:44-86,229-260;
 * R/wavefront_compute_affine2p.c:45-106,286-368; limits R/wavefront_compute.c:40-86;
 * which outputs exist R/wavefront_compute.c:440-485).  Returns 1 for a null step.
 */
/*
 * Single-component metrics.  gap-linear: R/wavefront_compute_linear.c:44-74,208-241 with limits
 * R/wavefront_compute.c:40-58; edit / indel: R/wavefront_compute_edit.c:44-100,374-418 (the same
 * recurrence with x = o = 1; indel has no mismatch term; these two never take a null step).
 */
static int compute_next_linear(oracle_ws_t* ws, int s, int plen, int tlen, int* err) {
  const int indel = (ws->metric == WFA_DIST_INDEL);
  wf_in_t mx = wf_fetch(ws, 0, indel ? -1 : s - ws->x);
  wf_in_t mo = wf_fetch(ws, 0, s - ws->o1);
  wf_slot(ws, 0, s)->exists = 0;
  if (mx.null && mo.null) {
    if (ef_required(ws, s) && ef_seed_null_step(ws, s)) *err = 1;
    return 1;
  }
  int lo = mo.lo - 1, hi = mo.hi + 1;
  if (!indel) { lo = MIN2(mx.lo, lo); hi = MAX2(mx.hi, hi); }
  {
    int alo = lo, ahi = hi;
    if (ef_required(ws, s)) ef_limits(ws, s, &alo, &ahi);
    if (wf_alloc_range(ws, 0, s, alo, ahi, lo, hi)) { *err = 1; return 0; }
  }
  mx = wf_fetch(ws, 0, indel ? -1 : s - ws->x);
  mo = wf_fetch(ws, 0, s - ws->o1);
  int32_t* om = ws->arena + wf_slot(ws, 0, s)->idx + (lo - wf_slot(ws, 0, s)->base);
  int k;
  g_m_offsets += (int64_t)hi - lo + 1; g_all_offsets += (int64_t)hi - lo + 1;
  for (k = lo; k <= hi; ++k) {
    const int32_t ins = wf_get(&mo, k - 1), del = wf_get(&mo, k + 1);
    int32_t mv = indel ? MAX2(del, ins + 1) : MAX2(del, MAX2(wf_get(&mx, k), ins) + 1);
    const uint32_t h = (uint32_t)mv;
    const uint32_t v = (uint32_t)(mv - k);
    if (h > (uint32_t)tlen) mv = OFFSET_NULL;
    if (v > (uint32_t)plen) mv = OFFSET_NULL;
    om[k - lo] = mv;
  }
  if (ef_required(ws, s)) ef_seed_computed(ws, wf_slot(ws, 0, s), s);
  wf_trim(ws, wf_slot(ws, 0, s), plen, tlen);
  return 0;
}

static int compute_next(oracle_ws_t* ws, int s, int plen, int tlen, int* err) {
  if (ws->ncomp == 1) return compute_next_linear(ws, s, plen, tlen, err);
  const int two = (ws->ncomp == 5);
  wf_in_t mx = wf_fetch(ws, 0, s - ws->x);
  wf_in_t mo1 = wf_fetch(ws, 0, s - ws->o1 - ws->e1);
  wf_in_t i1e = wf_fetch(ws, 1, s - ws->e1);
  wf_in_t d1e = wf_fetch(ws, 2, s - ws->e1);
  wf_in_t mo2 = wf_fetch(ws, 0, -1), i2e = mo2, d2e = mo2; /* null wavefronts */
  if (two) {
    mo2 = wf_fetch(ws, 0, s - ws->o2 - ws->e2);
    i2e = wf_fetch(ws, 3, s - ws->e2);
    d2e = wf_fetch(ws, 4, s - ws->e2);
  }
  int c;
  /* the slot is being overwritten (modular) or is fresh (full) */
  for (c = 0; c < ws->ncomp; ++c) wf_slot(ws, c, s)->exists = 0;
  if (mx.null && mo1.null && i1e.null && d1e.null && (!two || (mo2.null && i2e.null && d2e.null))) {
    if (ef_required(ws, s) && ef_seed_null_step(ws, s)) *err = 1;
    return 1;
  }
  int lo = mx.lo, hi = mx.hi;
  lo = MIN2(lo, mo1.lo - 1); hi = MAX2(hi, mo1.hi + 1);
  lo = MIN2(lo, i1e.lo + 1); hi = MAX2(hi, i1e.hi + 1);
  lo = MIN2(lo, d1e.lo - 1); hi = MAX2(hi, d1e.hi - 1);
  if (two) {
    lo = MIN2(lo, mo2.lo - 1); hi = MAX2(hi, mo2.hi + 1);
    lo = MIN2(lo, i2e.lo + 1); hi = MAX2(hi, i2e.hi + 1);
    lo = MIN2(lo, d2e.lo - 1); hi = MAX2(hi, d2e.hi - 1);
  }
  const int has_i1 = !mo1.null || !i1e.null;
  const int has_d1 = !mo1.null || !d1e.null;
  const int has_i2 = two && (!mo2.null || !i2e.null);
  const int has_d2 = two && (!mo2.null || !d2e.null);
  /* the reference delegates to the 1-piece kernel when all *2 inputs are null; the values
   * it would have produced for I2/D2 are then never stored, so the result is identical */
  {
    int alo = lo, ahi = hi;
    if (ef_required(ws, s)) ef_limits(ws, s, &alo, &ahi);
    if (wf_alloc_range(ws, 0, s, alo, ahi, lo, hi)) { *err = 1; return 0; }
  }
  if (has_i1 && wf_alloc(ws, 1, s, lo, hi)) { *err = 1; return 0; }
  if (has_d1 && wf_alloc(ws, 2, s, lo, hi)) { *err = 1; return 0; }
  if (has_i2 && wf_alloc(ws, 3, s, lo, hi)) { *err = 1; return 0; }
  if (has_d2 && wf_alloc(ws, 4, s, lo, hi)) { *err = 1; return 0; }
  /* arena may have moved: re-fetch input pointers */
  mx = wf_fetch(ws, 0, s - ws->x);
  mo1 = wf_fetch(ws, 0, s - ws->o1 - ws->e1);
  i1e = wf_fetch(ws, 1, s - ws->e1);
  d1e = wf_fetch(ws, 2, s - ws->e1);
  if (two) {
    mo2 = wf_fetch(ws, 0, s - ws->o2 - ws->e2);
    i2e = wf_fetch(ws, 3, s - ws->e2);
    d2e = wf_fetch(ws, 4, s - ws->e2);
  }
  int32_t* om = ws->arena + wf_slot(ws, 0, s)->idx + (lo - wf_slot(ws, 0, s)->base);
  int32_t* oi1 = has_i1 ? ws->arena + wf_slot(ws, 1, s)->idx : NULL;
  int32_t* od1 = has_d1 ? ws->arena + wf_slot(ws, 2, s)->idx : NULL;
  int32_t* oi2 = has_i2 ? ws->arena + wf_slot(ws, 3, s)->idx : NULL;
  int32_t* od2 = has_d2 ? ws->arena + wf_slot(ws, 4, s)->idx : NULL;
  int k;
  g_m_offsets += (int64_t)hi - lo + 1;
  g_all_offsets += ((int64_t)hi - lo + 1) * (1 + has_i1 + has_d1 + has_i2 + has_d2);
  for (k = lo; k <= hi; ++k) {
    const int32_t ins1 = MAX2(wf_get(&mo1, k - 1), wf_get(&i1e, k - 1)) + 1;
    const int32_t del1 = MAX2(wf_get(&mo1, k + 1), wf_get(&d1e, k + 1));
    int32_t ins = ins1, del = del1;
    if (oi1) oi1[k - lo] = ins1;
    if (od1) od1[k - lo] = del1;
    if (two) {
      const int32_t ins2 = MAX2(wf_get(&mo2, k - 1), wf_get(&i2e, k - 1)) + 1;
      const int32_t del2 = MAX2(wf_get(&mo2, k + 1), wf_get(&d2e, k + 1));
      if (oi2) oi2[k - lo] = ins2;
      if (od2) od2[k - lo] = del2;
      ins = MAX2(ins1, ins2);
      del = MAX2(del1, del2);
    }
    const int32_t misms = wf_get(&mx, k) + 1;
    int32_t mv = MAX2(del, MAX2(misms, ins));
    /* only M is clamped (R/wavefront_compute_affine.c:80-84) */
    const uint32_t h = (uint32_t)mv;
    const uint32_t v = (uint32_t)(mv - k);
    if (h > (uint32_t)tlen) mv = OFFSET_NULL;
    if (v > (uint32_t)plen) mv = OFFSET_NULL;
    om[k - lo] = mv;
  }
  if (ef_required(ws, s)) ef_seed_computed(ws, wf_slot(ws, 0, s), s);
  for (c = 0; c < ws->ncomp; ++c) {
    wf_t* w = wf_slot(ws, c, s);
    if (w->exists) wf_trim(ws, w, plen, tlen);
  }
  return 0;
}

typedef struct {
  int steps_wait;
  int have_max_sw; /* max_sw_score_k != DPMATRIX_DIAGONAL_NULL */
  int max_sw;
} heur_t;

/* R/wavefront_heuristic.c:161-172 (wf_heuristic_equate) */
static void heur_equate(wf_t* dst, const wf_t* src) {
  if (!dst->exists) return;
  if (src->lo > dst->lo) dst->lo = src->lo;
  if (src->hi < dst->hi) dst->hi = src->hi;
}

/*
 * Heuristic cut-off after extend (R/wavefront_heuristic.c:509-567), with wf-adaptive
 * (:176-192,232-293) and X-drop (:297-383).
 */
static int heur_cutoff(oracle_ws_t* ws, const wfa_hip_config_t* cfg, heur_t* hs, int s,
                       int plen, int tlen) {
  wf_t* m = wf_slot(ws, 0, s);
  if (!m->exists || m->lo > m->hi) return 0;
  --hs->steps_wait;
  const int lo_base = m->lo, hi_base = m->hi;
  const int32_t* off = ws->arena + m->idx;
  if (cfg->heuristic == WFA_HEUR_ADAPTIVE) {
    if (hs->steps_wait <= 0 && (hi_base - lo_base + 1) >= cfg->min_wavefront_length) {
      const int64_t n = (int64_t)hi_base - lo_base + 1;
      if (n > ws->tmp_cap) {
        free(ws->tmp);
        ws->tmp = (int32_t*)malloc((size_t)n * 2 * sizeof(int32_t));
        if (!ws->tmp) return -1;
        ws->tmp_cap = n * 2;
      }
      int32_t* dist = ws->tmp;
      int k, min_d = MAX2(plen, tlen);
      for (k = lo_base; k <= hi_base; ++k) {
        const int32_t o = off[k - m->base];
        const int left_v = plen - (o - k);
        const int left_h = tlen - o;
        const int d = (o >= 0) ? MAX2(left_v, left_h) : -OFFSET_NULL;
        dist[k - lo_base] = d;
        min_d = MIN2(min_d, d);
      }
      const int thr = cfg->max_distance_threshold;
      const int ak = tlen - plen;
      const int top_limit = MIN2(ak, m->hi);
      int lo_red = m->lo;
      for (k = m->lo; k < top_limit; ++k) {
        if (dist[k - lo_base] - min_d <= thr) break;
        ++lo_red;
      }
      m->lo = lo_red;
      const int bottom_limit = MAX2(ak, m->lo);
      int hi_red = m->hi;
      for (k = m->hi; k > bottom_limit; --k) {
        if (dist[k - lo_base] - min_d <= thr) break;
        --hi_red;
      }
      m->hi = hi_red;
      hs->steps_wait = cfg->steps_between_cutoffs;
    }
  } else if (cfg->heuristic == WFA_HEUR_XDROP) {
    if (hs->steps_wait <= 0) {
      const int g = (ws->match != 0) ? -ws->match : -1; /* R/wavefront_heuristic.c:306-307 */
      int k, cmax = INT_MIN;
      const int64_t n = (int64_t)hi_base - lo_base + 1;
      if (n > ws->tmp_cap) {
        free(ws->tmp);
        ws->tmp = (int32_t*)malloc((size_t)n * 2 * sizeof(int32_t));
        if (!ws->tmp) return -1;
        ws->tmp_cap = n * 2;
      }
      int32_t* sw = ws->tmp;
      for (k = lo_base; k <= hi_base; ++k) {
        const int32_t o = off[k - m->base];
        if (o < 0) continue;
        const int v = o - k, h = o;
        const int sc = (g * (v + h) - s) / 2; /* C truncating division */
        sw[k - lo_base] = sc;
        if (cmax < sc) cmax = sc;
      }
      if (hs->have_max_sw) {
        const int max_sw = hs->max_sw;
        for (k = m->lo; k <= m->hi; ++k) {
          if (off[k - m->base] < 0) continue;
          if (max_sw - sw[k - lo_base] < cfg->xdrop) break;
        }
        m->lo = k;
        for (k = m->hi; k >= m->lo; --k) {
          if (off[k - m->base] < 0) continue;
          if (max_sw - sw[k - lo_base] < cfg->xdrop) break;
        }
        m->hi = k;
        if (cmax > hs->max_sw) hs->max_sw = cmax;
      } else {
        hs->max_sw = cmax;
        hs->have_max_sw = 1;
      }
      hs->steps_wait = cfg->steps_between_cutoffs;
    }
  }
  if (lo_base == m->lo && hi_base == m->hi) return 0;
  int c;
  for (c = 1; c < ws->ncomp; ++c) heur_equate(wf_slot(ws, c, s), m);
  return 0;
}

/* candidate readers of the backtrace (R/wavefront_backtrace.c:64-219): value only if the
 * stored wavefront exists and k lies in its stored [lo,hi]; (offset<<4)|type otherwise NULL */
static inline int64_t bt_cand(oracle_ws_t* ws, int c, int s, int k, int add, int type) {
  if (s < 0) return OFFSET_NULL;
  const wf_t* w = wf_slot(ws, c, s);
  if (!w->exists || k < w->lo || k > w->hi) return OFFSET_NULL;
  const int32_t o = ws->arena[w->idx + (k - w->base)];
  return (((int64_t)(o + add)) << 4) | type;
}

typedef struct {
  uint8_t* buf; /* written right-to-left */
  int64_t begin; /* index of the first valid op */
  int64_t end;
} ops_t;

static inline void ops_push(ops_t* ops, char c, int n) {
  while (n-- > 0) ops->buf[--ops->begin] = (uint8_t)c;
}

/* R/wavefront_backtrace.c:320-529 (wavefront_backtrace_affine); comp_begin / comp_end (0=M 1=I1 2=D1 3=I2 4=D2) are M
 * for an ordinary alignment, another component for the halves of a BiWFA split (R/wavefront_bialign.c:581-658) */
static void backtrace_c(oracle_ws_t* ws, int plen, int tlen, int end_s, int end_k, int32_t end_off,
                        ops_t* ops, int comp_begin, int comp_end);
static void backtrace(oracle_ws_t* ws, int plen, int tlen, int end_s, int end_k, int32_t end_off,
                      ops_t* ops) {
  backtrace_c(ws, plen, tlen, end_s, end_k, end_off, ops, 0, 0);
}
static void backtrace_c(oracle_ws_t* ws, int plen, int tlen, int end_s, int end_k, int32_t end_off,
                        ops_t* ops, int comp_begin, int comp_end) {
  (void)comp_begin;
  enum { BT_I1_OPEN = 1, BT_I1_EXT, BT_I2_OPEN, BT_I2_EXT, BT_D1_OPEN, BT_D1_EXT, BT_D2_OPEN,
         BT_D2_EXT, BT_M };
  const int two = (ws->ncomp == 5);
  if (ws->ncomp == 1) {
    /* R/wavefront_backtrace.c:223-319 (wavefront_backtrace_linear) */
    int s = end_s, k = end_k;
    int32_t offset = end_off;
    int h = offset, v = offset - k;
    if (v < plen) ops_push(ops, 'D', plen - v);
    if (h < tlen) ops_push(ops, 'I', tlen - h);
    while (v > 0 && h > 0 && s > 0) {
      const int s_x = s - ws->x, s_o = s - ws->o1;
      int64_t best = (ws->metric != WFA_DIST_INDEL) ? bt_cand(ws, 0, s_x, k, 1, BT_M) : OFFSET_NULL;
      best = MAX2(best, bt_cand(ws, 0, s_o, k - 1, 1, BT_I1_OPEN));
      best = MAX2(best, bt_cand(ws, 0, s_o, k + 1, 0, BT_D1_OPEN));
      if (best < 0) break;
      const int32_t src = (int32_t)(best >> 4);
      ops_push(ops, 'M', offset - src);
      offset = src;
      v = offset - k; h = offset;
      if (v <= 0 || h <= 0) break;
      const int type = (int)(best & 0xF);
      if (type == BT_M) { s = s_x; ops_push(ops, 'X', 1); --offset; }
      else if (type == BT_I1_OPEN) { s = s_o; ops_push(ops, 'I', 1); --k; --offset; }
      else { s = s_o; ops_push(ops, 'D', 1); ++k; }
      v = offset - k; h = offset;
    }
    if (v > 0 && h > 0) { const int n = MIN2(v, h); ops_push(ops, 'M', n); v -= n; h -= n; }
    ops_push(ops, 'D', v > 0 ? v : 0);
    ops_push(ops, 'I', h > 0 ? h : 0);
    return;
  }
  int comp = comp_end; /* 0=M 1=I1 2=D1 3=I2 4=D2 */
  int s = end_s, k = end_k;
  int32_t offset = end_off;
  int h = offset, v = offset - k;
  if (comp_end == 0) { /* R/wavefront_backtrace.c:346-355 */
    if (v < plen) ops_push(ops, 'D', plen - v);
    if (h < tlen) ops_push(ops, 'I', tlen - h);
  }
  while (v > 0 && h > 0 && s > 0) {
    const int s_x = s - ws->x;
    const int s_o1 = s - ws->o1 - ws->e1, s_e1 = s - ws->e1;
    const int s_o2 = s - ws->o2 - ws->e2, s_e2 = s - ws->e2;
    int64_t best;
    if (comp == 0) {
      best = bt_cand(ws, 0, s_x, k, 1, BT_M);
      best = MAX2(best, bt_cand(ws, 0, s_o1, k - 1, 1, BT_I1_OPEN));
      best = MAX2(best, bt_cand(ws, 1, s_e1, k - 1, 1, BT_I1_EXT));
      best = MAX2(best, bt_cand(ws, 0, s_o1, k + 1, 0, BT_D1_OPEN));
      best = MAX2(best, bt_cand(ws, 2, s_e1, k + 1, 0, BT_D1_EXT));
      if (two) {
        best = MAX2(best, bt_cand(ws, 0, s_o2, k - 1, 1, BT_I2_OPEN));
        best = MAX2(best, bt_cand(ws, 3, s_e2, k - 1, 1, BT_I2_EXT));
        best = MAX2(best, bt_cand(ws, 0, s_o2, k + 1, 0, BT_D2_OPEN));
        best = MAX2(best, bt_cand(ws, 4, s_e2, k + 1, 0, BT_D2_EXT));
      }
    } else if (comp == 1) {
      best = MAX2(bt_cand(ws, 0, s_o1, k - 1, 1, BT_I1_OPEN), bt_cand(ws, 1, s_e1, k - 1, 1, BT_I1_EXT));
    } else if (comp == 3) {
      best = MAX2(bt_cand(ws, 0, s_o2, k - 1, 1, BT_I2_OPEN), bt_cand(ws, 3, s_e2, k - 1, 1, BT_I2_EXT));
    } else if (comp == 2) {
      best = MAX2(bt_cand(ws, 0, s_o1, k + 1, 0, BT_D1_OPEN), bt_cand(ws, 2, s_e1, k + 1, 0, BT_D1_EXT));
    } else {
      best = MAX2(bt_cand(ws, 0, s_o2, k + 1, 0, BT_D2_OPEN), bt_cand(ws, 4, s_e2, k + 1, 0, BT_D2_EXT));
    }
    if (best < 0) break;
    if (comp == 0) {
      const int32_t src = (int32_t)(best >> 4);
      ops_push(ops, 'M', offset - src);
      offset = src;
      v = offset - k;
      h = offset;
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
    if (type == BT_M) {
      ops_push(ops, 'X', 1);
      --offset;
    } else if (type <= BT_I2_EXT) {
      ops_push(ops, 'I', 1);
      --k;
      --offset;
    } else {
      ops_push(ops, 'D', 1);
      ++k;
    }
    v = offset - k;
    h = offset;
  }
  if (comp == 0) {
    if (v > 0 && h > 0) {
      const int n = MIN2(v, h);
      ops_push(ops, 'M', n);
      v -= n;
      h -= n;
    }
    ops_push(ops, 'D', v > 0 ? v : 0);
    ops_push(ops, 'I', h > 0 ? h : 0);
  }
}

/* R/wavefront_compute.c:108-120 + R/wavefront_penalties.h:73 */
static int classic_score(const oracle_ws_t* ws, int v, int h, int s) {
  if (ws->metric <= WFA_DIST_EDIT) return s; /* distances are reported as they are */
  if (ws->match == 0) return -s;
  return ((-ws->match) * (v + h) - s) / 2;
}

/*
 * One alignment (R/wavefront_align.c:212-240 → R/wavefront_unialign.c:54-94,241-273,147-237).
 * ops (nullable) must hold plen+tlen bytes; the op string ends at ops+plen+tlen and starts
 * at *ops_begin.  Returns 0, or -1 when out of memory.
 */
static int align_one(oracle_ws_t* ws, const wfa_hip_config_t* cfg, const uint8_t* P, int plen,
                     const uint8_t* T, int tlen, int32_t* out_score, int32_t* out_status,
                     uint8_t* ops_buf, int64_t* ops_begin, int32_t* ops_len) {
  const int endsfree = (cfg->span == WFA_SPAN_ENDSFREE);
  const int full = (cfg->scope == WFA_SCOPE_FULL);
  /* Q13: free begins are only initialised for ends-free spans */
  const int pbf = (endsfree && ws->match == 0) ? cfg->pattern_begin_free : 0;
  const int tbf = (endsfree && ws->match == 0) ? cfg->text_begin_free : 0;
  const int64_t max_steps = (cfg->max_steps <= 0) ? INT_MAX : cfg->max_steps;
  const int wc = cfg->wildcard;
  int err = 0;
  /* R/wavefront_compute.c:124-139 (wavefront_compute_endsfree_required, without its per-score test) */
  ws->ef_pbf = cfg->pattern_begin_free; ws->ef_tbf = cfg->text_begin_free;
  ws->ef_seed = (ws->match != 0 && endsfree && (ws->ef_pbf != 0 || ws->ef_tbf != 0));
  ws->modular = !full;
  ws->arena_used = 0;
  if (ws->modular) {
    ws->slot_stride = (int64_t)plen + tlen + 8;
    if (ws_reserve_scores(ws, ws->scope)) return -1;
    if (ws_reserve_arena(ws, ws->slot_stride * 5 * ws->scope)) return -1;
    int c, i;
    for (c = 0; c < 5; ++c) for (i = 0; i < ws->scope; ++i) ws->wf[c][i].exists = 0;
  } else {
    if (ws_reserve_scores(ws, 64)) return -1;
  }
  *ops_len = 0;
  if (ops_begin) *ops_begin = 0;
  /* wavefront 0 (R/wavefront_aligner.c:251-310) */
  if (wf_alloc(ws, 0, 0, -pbf, tbf)) return -1;
  {
    int32_t* o = ws->arena + ws->wf[0][0].idx;
    int k;
    for (k = -pbf; k <= tbf; ++k) o[k + pbf] = (k > 0) ? k : 0;
    int c;
    for (c = 1; c < 5; ++c) ws->wf[c][0].exists = 0;
  }
  heur_t hs;
  hs.steps_wait = cfg->steps_between_cutoffs; /* R/wavefront_heuristic.c:114-121 */
  hs.have_max_sw = 0;
  hs.max_sw = 0;
  int null_steps = 0;
  int s = 0;
  int end_reason = 0, end_k = 0;
  int32_t end_off = OFFSET_NULL;
  for (;;) {
    /* ---- extend + termination + cut-off (R/wavefront_extend.c:90-125,263-297) ---- */
    wf_t* m = wf_slot(ws, 0, s);
    if (!m->exists) {
      if (null_steps > ws->scope) { end_reason = END_UNREACHABLE; break; }
    } else {
      int32_t* off = ws->arena + m->idx;
      int k;
      for (k = m->lo; k <= m->hi; ++k) {
        int32_t o = off[k - m->base];
        if (o == OFFSET_NULL) continue;
        int v = o - k, h = o;
        /* R/wavefront_extend_kernels.c:64-88: byte-equal run, stopped by the sentinels */
        if (wc < 0) {
          while (v < plen && h < tlen && P[v] == T[h]) { ++v; ++h; }
        } else {
          while (v < plen && h < tlen && (P[v] == T[h] || P[v] == wc || T[h] == wc)) { ++v; ++h; }
        }
        g_bases += h - o + 1;
        o = h;
        off[k - m->base] = o;
        if (endsfree) {
          /* R/wavefront_termination.c:115-162, tested after each k in ascending order */
          if ((h >= tlen && plen - v <= cfg->pattern_end_free) ||
              (v >= plen && tlen - h <= cfg->text_end_free)) {
            end_reason = END_REACHED; end_k = k; end_off = o;
            break;
          }
        }
      }
      if (!endsfree) {
        /* R/wavefront_termination.c:37-61 */
        const int ak = tlen - plen;
        if (m->lo <= ak && ak <= m->hi && off[ak - m->base] >= tlen) {
          end_reason = END_REACHED; end_k = ak; end_off = tlen;
        }
      }
      if (end_reason) break;
      if (cfg->heuristic != WFA_HEUR_NONE) {
        if (heur_cutoff(ws, cfg, &hs, s, plen, tlen)) return -1;
      }
    }
    /* ---- next score (R/wavefront_unialign.c:262-265) ---- */
    ++s;
    if (!ws->modular && ws_reserve_scores(ws, (int64_t)s + 1)) return -1;
    if (compute_next(ws, s, plen, tlen, &err)) ++null_steps; else null_steps = 0;
    if (err) return -1;
    if (s >= max_steps) {
      /* R/wavefront_unialign.c:102-107 */
      *out_status = WFA_STATUS_MAX_STEPS_REACHED;
      *out_score = (int32_t)(-max_steps);
      return 0;
    }
  }
  /* ---- finish (R/wavefront_unialign.c:147-237) ---- */
  if (!full) {
    if (end_reason == END_REACHED) {
      *out_score = classic_score(ws, plen, tlen, s);
      *out_status = WFA_STATUS_COMPLETED;
    } else {
      /* the reference evaluates the score at its (unset) end position; with match==0 that
       * is -s; with match<0 it is computed from k=INT_MAX, offset=NULL (wrapping) */
      const int32_t k = INT_MAX;
      const int32_t ev = (int32_t)((uint32_t)OFFSET_NULL - (uint32_t)k);
      *out_score = (ws->match == 0) ? -s : (int32_t)(((int64_t)(-ws->match) * ((int64_t)ev + OFFSET_NULL) - s) / 2);
      *out_status = WFA_STATUS_PARTIAL;
    }
    return 0;
  }
  if (end_reason == END_REACHED) {
    ops_t ops;
    ops.buf = ops_buf;
    ops.end = (int64_t)plen + tlen;
    ops.begin = ops.end;
    backtrace(ws, plen, tlen, s, end_k, end_off, &ops);
    *ops_begin = ops.begin;
    *ops_len = (int32_t)(ops.end - ops.begin);
    *out_score = classic_score(ws, end_off - end_k, end_off, s);
    *out_status = WFA_STATUS_COMPLETED;
  } else {
    /* no backtrace; maxtrim of the empty CIGAR clears it (R/alignment/cigar.c:473-613) */
    *out_score = INT32_MIN;
    *out_status = WFA_STATUS_PARTIAL;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * BiWFA (memory_mode "biwfa" = wavefront_memory_ultralow), scope=full: R/wavefront_bialign.c.
 * Two score-only aligners walk towards each other (forward on the sequences, reverse on the reversed
 * sequences), the first overlap of their wavefronts gives a breakpoint (R/wavefront_bialign.c:189-395), the
 * two halves are aligned recursively with the breakpoint's component as end / begin component, and a half
 * whose score is <= 250 is aligned by the ordinary algorithm (R/wavefront_bialign.c:155-188, 581-607).
 * ------------------------------------------------------------------------------------------------ */
#define BI_FALLBACK_MIN_SCORE 250  /* R/wavefront_bialign.c:48 */
#define BI_FALLBACK_MIN_LENGTH 100 /* :49 */
#define BI_RECOVERY_MIN_SCORE 500  /* :50 */
/* internal status values (R/wfa.h:52-55) */
#define BI_OK (-1)
#define BI_END_REACHED (-2)
#define BI_END_UNREACHABLE (-3)

/* a window of the two sequences, read forwards or backwards (R/wavefront_sequences.c:275-310) */
typedef struct {
  const uint8_t* P; const uint8_t* T;
  int pbeg, pend, tbeg, tend;
  int reverse;
  int wc; /* wildcard byte or -1 */
} bi_view_t;

static inline int bi_match(const bi_view_t* w, int v, int h) {
  const uint8_t pc = w->reverse ? w->P[w->pend - 1 - v] : w->P[w->pbeg + v];
  const uint8_t tc = w->reverse ? w->T[w->tend - 1 - h] : w->T[w->tbeg + h];
  return pc == tc || (w->wc >= 0 && (pc == w->wc || tc == w->wc));
}

/* one unidirectional aligner (forward, reverse or base) */
typedef struct {
  oracle_ws_t ws;
  bi_view_t view;
  int plen, tlen;
  int comp_begin, comp_end;
  int null_steps;
  int status;       /* BI_OK while running, BI_END_REACHED, BI_END_UNREACHABLE */
  int status_score; /* align_status.score */
  int end_k; int32_t end_off;
  heur_t hs;        /* the forward / reverse aligners inherit the heuristic (R/wavefront_bialigner.c:53,161-166); the base aligner does not */
} bi_uni_t;

/* R/wavefront_aligner.c:251-417 with a begin component (:329-390): wavefront 0 is the cell (k=0, offset 0) of that component */
static int bi_uni_init(bi_uni_t* u, const bi_view_t* view, int comp_begin, int comp_end, int modular) {
  oracle_ws_t* ws = &u->ws;
  u->view = *view;
  u->plen = view->pend - view->pbeg;
  u->tlen = view->tend - view->tbeg;
  u->comp_begin = comp_begin; u->comp_end = comp_end;
  u->null_steps = 0; u->status = BI_OK; u->status_score = 0; u->end_k = 0; u->end_off = OFFSET_NULL;
  u->hs.steps_wait = 0; u->hs.have_max_sw = 0; u->hs.max_sw = 0;   /* (set by the breakpoint search: R/wavefront_heuristic.c:114-121 at every init) */
  ws->modular = modular;
  ws->arena_used = 0;
  ws->ef_seed = 0;   /* (BiWFA has no free ends: R/wavefront_align.c:60-75) */
  if (modular) {
    ws->slot_stride = (int64_t)u->plen + u->tlen + 8;
    if (ws_reserve_scores(ws, ws->scope)) return -1;
    if (ws_reserve_arena(ws, ws->slot_stride * 5 * ws->scope)) return -1;
    int c, i;
    for (c = 0; c < 5; ++c) for (i = 0; i < ws->scope; ++i) ws->wf[c][i].exists = 0;
  } else {
    if (ws_reserve_scores(ws, 64)) return -1;
    int c;
    for (c = 0; c < 5; ++c) ws->wf[c][0].exists = 0;
  }
  if (wf_alloc(ws, comp_begin, 0, 0, 0)) return -1;
  ws->arena[ws->wf[comp_begin][0].idx] = 0;
  return 0;
}

/* R/wavefront_termination.c:37-113 (end2end, any end component).  Only reached when M[s] exists
 * (R/wavefront_extend.c:90-125: the test sits behind the `mwavefront == NULL` return). */
static int bi_terminated(bi_uni_t* u, int s) {
  const wf_t* w = wf_slot(&u->ws, u->comp_end, s);
  const int ak = u->tlen - u->plen;
  if (!w->exists || w->lo > ak || ak > w->hi) return 0;
  if (u->ws.arena[w->idx + (ak - w->base)] < u->tlen) return 0;
  u->end_k = ak; u->end_off = u->tlen;
  return 1;
}

/* R/wavefront_extend.c:90-125 / :178-214 (wavefront_extend_end2end[_max]): extend M[s], test the end, then the heuristic cut-off
 * of the forward / reverse aligners (hcfg != NULL: R/wavefront_extend.c:117-123,206-212 -> R/wavefront_heuristic.c:509-567).
 * Returns 1 when the aligner is done (status set).  *max_ak (nullable) = largest antidiagonal 2*offset - k reached. */
static int bi_extend(bi_uni_t* u, int s, int* max_ak, const wfa_hip_config_t* hcfg) {
  oracle_ws_t* ws = &u->ws;
  wf_t* m = wf_slot(ws, 0, s);
  if (max_ak) *max_ak = 0;
  if (!m->exists) {
    if (u->null_steps > ws->scope) { u->status = BI_END_UNREACHABLE; u->status_score = s; return 1; }
    return 0;
  }
  int32_t* off = ws->arena + m->idx;
  int k, best = 0;
  for (k = m->lo; k <= m->hi; ++k) {
    int32_t o = off[k - m->base];
    if (o == OFFSET_NULL) continue;
    int v = o - k, h = o;
    while (v < u->plen && h < u->tlen && bi_match(&u->view, v, h)) { ++v; ++h; }
    off[k - m->base] = h;
    const int ad = 2 * h - k;
    if (best < ad) best = ad;
  }
  if (bi_terminated(u, s)) { u->status = BI_END_REACHED; u->status_score = s; return 1; }
  if (hcfg && hcfg->heuristic != WFA_HEUR_NONE) {
    if (heur_cutoff(ws, hcfg, &u->hs, s, u->plen, u->tlen)) { u->status = BI_END_UNREACHABLE; u->status_score = s; return 1; }   /* (allocation failure only) */
  }
  if (max_ak) *max_ak = best;
  return 0;
}

static int bi_compute(bi_uni_t* u, int s) {
  int err = 0;
  if (!u->ws.modular && ws_reserve_scores(&u->ws, (int64_t)s + 1)) return -1;
  if (compute_next(&u->ws, s, u->plen, u->tlen, &err)) ++u->null_steps; else u->null_steps = 0;
  return err ? -1 : 0;
}

typedef struct {
  int score, score_forward, score_reverse;
  int k_forward, k_reverse;
  int32_t offset_forward, offset_reverse;
  int component;
} bi_breakpoint_t;

/* R/wavefront_bialign.c:189-252 (indel2indel) and :253-311 (m2m): wavefront `c` of aligner 0 at score_0 against the same
 * component of aligner 1 at score_1; the first diagonal (ascending k of aligner 0) where the two offsets meet wins */
static void bi_breakpoint_cc(const bi_uni_t* u0, const bi_uni_t* u1, int forward, int score_0, int score_1,
                             const wf_t* w0, const wf_t* w1, int c, bi_breakpoint_t* bp) {
  const int plen = u0->plen, tlen = u0->tlen;
  const int gap_open = (c == 0) ? 0 : ((c == 1 || c == 2) ? u0->ws.o1 : u0->ws.o2);
  const int lo_0 = w0->lo, hi_0 = w0->hi;
  const int lo_1 = tlen - plen - w1->hi, hi_1 = tlen - plen - w1->lo; /* WAVEFRONT_K_INVERSE */
  if (hi_1 < lo_0 || hi_0 < lo_1) return;
  const int min_hi = MIN2(hi_0, hi_1), max_lo = MAX2(lo_0, lo_1);
  int k_0;
  for (k_0 = max_lo; k_0 <= min_hi; ++k_0) {
    const int k_1 = tlen - plen - k_0;
    const int32_t o0 = u0->ws.arena[w0->idx + (k_0 - w0->base)];
    const int32_t o1 = u1->ws.arena[w1->idx + (k_1 - w1->base)];
    if ((int64_t)o0 + o1 >= tlen && score_0 + score_1 - gap_open < bp->score) {
      if (c != 0) { /* out-of-bounds I/D offsets are kept by compute-next: skip them (R/wavefront_bialign.c:222-226,236-240) */
        const int kk = forward ? k_0 : k_1;
        const int32_t oo = forward ? o0 : o1;
        if (oo - kk > plen || oo > tlen) continue;
      }
      if (forward) {
        bp->score_forward = score_0; bp->score_reverse = score_1;
        bp->k_forward = k_0; bp->k_reverse = k_1; bp->offset_forward = o0; bp->offset_reverse = o1;
      } else {
        bp->score_forward = score_1; bp->score_reverse = score_0;
        bp->k_forward = k_1; bp->k_reverse = k_0; bp->offset_forward = o1; bp->offset_reverse = o0;
      }
      bp->score = score_0 + score_1 - gap_open;
      bp->component = c;
      return;
    }
  }
}

/* R/wavefront_bialign.c:315-395 (wavefront_bialign_overlap) */
static void bi_overlap(const bi_uni_t* u0, const bi_uni_t* u1, int score_0, int score_1, int forward, bi_breakpoint_t* bp) {
  const oracle_ws_t* ws0 = &u0->ws;
  const oracle_ws_t* ws1 = &u1->ws;
  const int scope = ws0->scope;
  const wf_t* m0 = &ws0->wf[0][score_0 % scope];
  if (!m0->exists) return;
  int i;
  for (i = 0; i < scope; ++i) {
    const int score_i = score_1 - i;
    if (score_i < 0) break;
    const int mod_i = score_i % scope, mod_0 = score_0 % scope;
    if (ws0->ncomp == 5) {
      if (score_0 + score_i - ws0->o2 >= bp->score) continue;
      if (ws0->wf[4][mod_0].exists && ws1->wf[4][mod_i].exists)
        bi_breakpoint_cc(u0, u1, forward, score_0, score_i, &ws0->wf[4][mod_0], &ws1->wf[4][mod_i], 4, bp);
      if (ws0->wf[3][mod_0].exists && ws1->wf[3][mod_i].exists)
        bi_breakpoint_cc(u0, u1, forward, score_0, score_i, &ws0->wf[3][mod_0], &ws1->wf[3][mod_i], 3, bp);
    }
    if (ws0->ncomp >= 3) {
      if (score_0 + score_i - ws0->o1 >= bp->score) continue;
      if (ws0->wf[2][mod_0].exists && ws1->wf[2][mod_i].exists)
        bi_breakpoint_cc(u0, u1, forward, score_0, score_i, &ws0->wf[2][mod_0], &ws1->wf[2][mod_i], 2, bp);
      if (ws0->wf[1][mod_0].exists && ws1->wf[1][mod_i].exists)
        bi_breakpoint_cc(u0, u1, forward, score_0, score_i, &ws0->wf[1][mod_0], &ws1->wf[1][mod_i], 1, bp);
    }
    if (score_0 + score_i >= bp->score) continue;
    if (ws1->wf[0][mod_i].exists)
      bi_breakpoint_cc(u0, u1, forward, score_0, score_i, m0, &ws1->wf[0][mod_i], 0, bp);
  }
}

typedef struct {
  bi_uni_t fwd, rev, base;
  const uint8_t* P; const uint8_t* T;
  int wc;
  int64_t max_steps;
  const wfa_hip_config_t* cfg; /* heuristic of the forward / reverse aligners */
  uint8_t* ops; /* appended forwards (R/alignment/cigar.c:125-136) */
  int64_t ops_len;
} bi_ctx_t;

/* R/wavefront_bialign.c:411-519 (wavefront_bialign_find_breakpoint); returns BI_OK, another BI_* status,
 * WFA_STATUS_MAX_STEPS_REACHED, or WFA_STATUS_OOM for an allocation failure */
static int bi_find_breakpoint(bi_ctx_t* cx, int pbeg, int pend, int tbeg, int tend, int comp_begin, int comp_end,
                              bi_breakpoint_t* bp) {
  bi_uni_t* f = &cx->fwd;
  bi_uni_t* r = &cx->rev;
  bi_view_t view = {cx->P, cx->T, pbeg, pend, tbeg, tend, 0, cx->wc};
  if (bi_uni_init(f, &view, comp_begin, comp_end, 1)) return WFA_STATUS_OOM;
  view.reverse = 1;
  if (bi_uni_init(r, &view, comp_end, comp_begin, 1)) return WFA_STATUS_OOM;
  const int plen = pend - pbeg, tlen = tend - tbeg;
  const int max_antidiagonal = plen + tlen - 1;
  int score_f = 0, score_r = 0, f_max_ak = 0, r_max_ak = 0, max_ak = 0;
  bp->score = INT_MAX;
  f->hs.steps_wait = r->hs.steps_wait = cx->cfg->steps_between_cutoffs;   /* R/wavefront_heuristic.c:114-121 (wavefront_heuristic_clear at every unialign init) */
  if (bi_extend(f, 0, &f_max_ak, cx->cfg)) return f->status;
  if (bi_extend(r, 0, &r_max_ak, cx->cfg)) return r->status;
  int last_forward = 0;
  for (;;) {
    if (f_max_ak + r_max_ak >= max_antidiagonal) break;
    ++score_f;
    if (bi_compute(f, score_f)) return WFA_STATUS_OOM;
    const int qf = bi_extend(f, score_f, &max_ak, cx->cfg);
    if (f_max_ak < max_ak) f_max_ak = max_ak;
    last_forward = 1;
    if (qf) return f->status;
    if (f_max_ak + r_max_ak >= max_antidiagonal) break;
    ++score_r;
    if (bi_compute(r, score_r)) return WFA_STATUS_OOM;
    const int qr = bi_extend(r, score_r, &max_ak, cx->cfg);
    if (r_max_ak < max_ak) r_max_ak = max_ak;
    last_forward = 0;
    if (qr) return r->status;
    if ((int64_t)score_r + score_f >= cx->max_steps) return WFA_STATUS_MAX_STEPS_REACHED;
  }
  const int scope = f->ws.scope;
  const int gap_opening = (f->ws.ncomp == 3) ? f->ws.o1 : (f->ws.ncomp == 5) ? MAX2(f->ws.o1, f->ws.o2) : 0;
  for (;;) {
    if (last_forward) {
      const int min_score_reverse = (score_r > scope - 1) ? score_r - (scope - 1) : 0;
      if (score_f + min_score_reverse - gap_opening >= bp->score) break;
      bi_overlap(f, r, score_f, score_r, 1, bp);
      ++score_r;
      if (bi_compute(r, score_r)) return WFA_STATUS_OOM;
      if (bi_extend(r, score_r, NULL, cx->cfg)) return r->status;
    }
    const int min_score_forward = (score_f > scope - 1) ? score_f - (scope - 1) : 0;
    if (min_score_forward + score_r - gap_opening >= bp->score) break;
    bi_overlap(r, f, score_r, score_f, 0, bp);
    ++score_f;
    if (bi_compute(f, score_f)) return WFA_STATUS_OOM;
    if (bi_extend(f, score_f, NULL, cx->cfg)) return f->status;
    if ((int64_t)score_r + score_f >= cx->max_steps) return WFA_STATUS_MAX_STEPS_REACHED;
    last_forward = 1;
  }
  return BI_OK;
}

/* R/wavefront_bialign.c:155-188 (wavefront_bialign_base): the ordinary algorithm on the window, full history, no
 * heuristic, begin / end component as given; the op string is appended.  `endsfree`: the extension / termination form
 * (R/wavefront_unialign.c:84-88: only the top-level call passes pywfa's ends-free span, with all free ends 0). */
static int bi_base(bi_ctx_t* cx, int pbeg, int pend, int tbeg, int tend, int comp_begin, int comp_end, int endsfree) {
  bi_uni_t* u = &cx->base;
  bi_view_t view = {cx->P, cx->T, pbeg, pend, tbeg, tend, 0, cx->wc};
  if (bi_uni_init(u, &view, comp_begin, comp_end, 0)) return WFA_STATUS_OOM;
  int s = 0, finished = 0;
  for (;;) {
    if (endsfree) {
      /* R/wavefront_extend.c:263-297 with R/wavefront_termination.c:115-162, free ends 0: M only */
      wf_t* m = wf_slot(&u->ws, 0, s);
      if (!m->exists) {
        if (u->null_steps > u->ws.scope) { u->status = BI_END_UNREACHABLE; finished = 1; }
      } else {
        int32_t* off = u->ws.arena + m->idx;
        int k;
        for (k = m->lo; k <= m->hi && !finished; ++k) {
          int32_t o = off[k - m->base];
          if (o == OFFSET_NULL) continue;
          int v = o - k, h = o;
          while (v < u->plen && h < u->tlen && bi_match(&u->view, v, h)) { ++v; ++h; }
          off[k - m->base] = h;
          if ((h >= u->tlen && u->plen - v <= 0) || (v >= u->plen && u->tlen - h <= 0)) {
            u->status = BI_END_REACHED; u->end_k = k; u->end_off = h; finished = 1;
          }
        }
      }
    } else {
      finished = bi_extend(u, s, NULL, NULL);   /* the base aligner has no heuristic (R/wavefront_bialigner.c:66-68) */
    }
    if (finished) break;
    ++s;
    if (bi_compute(u, s)) return WFA_STATUS_OOM;
    if (s >= cx->max_steps) return WFA_STATUS_UNATTAINABLE; /* base status MAX_STEPS != COMPLETED (R/wavefront_bialign.c:182-187) */
  }
  if (u->status != BI_END_REACHED) return WFA_STATUS_UNATTAINABLE;
  /* backtrace into a scratch buffer (right to left), then append forwards */
  const int64_t cap = (int64_t)u->plen + u->tlen;
  uint8_t* tmp = (uint8_t*)malloc((size_t)(cap > 0 ? cap : 1));
  if (!tmp) return WFA_STATUS_OOM;
  ops_t ops;
  ops.buf = tmp; ops.end = cap; ops.begin = cap;
  backtrace_c(&u->ws, u->plen, u->tlen, s, u->end_k, u->end_off, &ops, comp_begin, comp_end);
  memcpy(cx->ops + cx->ops_len, tmp + ops.begin, (size_t)(ops.end - ops.begin));
  cx->ops_len += ops.end - ops.begin;
  free(tmp);
  return BI_OK;
}

/* R/wavefront_bialign.c:581-658 (wavefront_bialign_alignment); *bp_score receives the breakpoint's score at this level */
static int bi_alignment(bi_ctx_t* cx, int pbeg, int pend, int tbeg, int tend, int comp_begin, int comp_end,
                        int score_remaining, int level, int endsfree_form, int* bp_score) {
  const int plen = pend - pbeg, tlen = tend - tbeg;
  if (tlen == 0) { memset(cx->ops + cx->ops_len, 'D', (size_t)plen); cx->ops_len += plen; return BI_OK; }
  if (plen == 0) { memset(cx->ops + cx->ops_len, 'I', (size_t)tlen); cx->ops_len += tlen; return BI_OK; }
  if (score_remaining <= BI_FALLBACK_MIN_SCORE) return bi_base(cx, pbeg, pend, tbeg, tend, comp_begin, comp_end, endsfree_form);
  bi_breakpoint_t bp;
  int st = bi_find_breakpoint(cx, pbeg, pend, tbeg, tend, comp_begin, comp_end, &bp);
  if (st != BI_OK) {
    /* R/wavefront_bialign.c:520-548 (wavefront_bialign_find_breakpoint_exception) */
    if (st == BI_END_REACHED) {
      const int reached = (cx->fwd.status == BI_END_REACHED) ? cx->fwd.status_score : cx->rev.status_score;
      if (reached <= BI_RECOVERY_MIN_SCORE) return bi_base(cx, pbeg, pend, tbeg, tend, comp_begin, comp_end, endsfree_form);
      return BI_END_UNREACHABLE;
    }
    return st;
  }
  const int bh = bp.offset_forward, bv = bp.offset_forward - bp.k_forward;
  st = bi_alignment(cx, pbeg, pbeg + bv, tbeg, tbeg + bh, comp_begin, bp.component, bp.score_forward, level + 1, 0, NULL);
  if (st != BI_OK) return st;
  st = bi_alignment(cx, pbeg + bv, pend, tbeg + bh, tend, bp.component, comp_end, bp.score_reverse, level + 1, 0, NULL);
  if (st != BI_OK) return st;
  if (bp_score) *bp_score = bp.score;
  return BI_OK;
}

/* R/wavefront_bialign.c:703-730 (wavefront_bialign, scope=full) for one pair.  Quirk reproduced (SURVEY.md Appendix B,
 * Q6): cigar->score is written only after a level-0 split (:651-656); when the top level is answered by the ordinary
 * algorithm (both sequences <= 100 bases, or one direction reaches the end before the wavefronts overlap) the score
 * stays at cigar_clear's INT32_MIN although the status is 0 and the op string is right. */
static int bi_align_one(bi_ctx_t* cx, const wfa_hip_config_t* cfg, const uint8_t* P, int plen, const uint8_t* T, int tlen,
                        int32_t* out_score, int32_t* out_status, uint8_t* ops_buf, int64_t* ops_begin, int32_t* ops_len) {
  cx->P = P; cx->T = T;
  cx->ops = ops_buf; cx->ops_len = 0;
  const int min_length = MAX2(plen, tlen) <= BI_FALLBACK_MIN_LENGTH;
  int bp_score = INT_MIN, have_bp = 0;
  int tmp_score = INT_MIN;
  int st = bi_alignment(cx, 0, plen, 0, tlen, 0, 0, min_length ? 0 : INT_MAX, 0, cfg->span == WFA_SPAN_ENDSFREE, &tmp_score);
  if (st == BI_OK && tmp_score != INT_MIN) { bp_score = tmp_score; have_bp = 1; }
  *ops_begin = 0;
  *ops_len = (int32_t)cx->ops_len;
  if (st == BI_OK) {
    *out_status = WFA_STATUS_COMPLETED;
    *out_score = have_bp ? classic_score(&cx->fwd.ws, plen, tlen, bp_score) : INT32_MIN;
  } else if (st == WFA_STATUS_MAX_STEPS_REACHED || st == WFA_STATUS_OOM) {
    *out_status = st; *out_score = INT32_MIN;
  } else {
    *out_status = -300; /* WF_STATUS_UNATTAINABLE */
    *out_score = INT32_MIN;
  }
  return (st == WFA_STATUS_OOM) ? -1 : 0;
}

/* R/wavefront_penalties.c:95-173 */
/* R/wavefront_bialign.c:662-702 (wavefront_bialign_compute_score) + the status translation of :703-730: the top-level
 * breakpoint search alone.  OK or END_REACHED (one direction crossed the whole matrix before any overlap): completed with
 * the breakpoint's / the reached score; anything else leaves the score at cigar_clear's INT32_MIN. */
static int bi_score_one(bi_ctx_t* cx, const uint8_t* P, int plen, const uint8_t* T, int tlen, int32_t* out_score, int32_t* out_status) {
  cx->P = P; cx->T = T;
  bi_breakpoint_t bp;
  const int st = bi_find_breakpoint(cx, 0, plen, 0, tlen, 0, 0, &bp);
  if (st == WFA_STATUS_OOM) return -1;
  if (st == BI_OK || st == BI_END_REACHED) {
    const int sc = (st == BI_OK) ? bp.score : ((cx->fwd.status == BI_END_REACHED) ? cx->fwd.status_score : cx->rev.status_score);
    *out_score = classic_score(&cx->fwd.ws, plen, tlen, sc);
    *out_status = WFA_STATUS_COMPLETED;
  } else {
    *out_score = INT32_MIN;
    *out_status = (st == WFA_STATUS_MAX_STEPS_REACHED) ? WFA_STATUS_MAX_STEPS_REACHED : -300;
  }
  return 0;
}

static int ws_set_penalties(oracle_ws_t* ws, const wfa_hip_config_t* cfg) {
  ws->metric = cfg->distance;
  if (cfg->distance == WFA_DIST_INDEL || cfg->distance == WFA_DIST_EDIT) {
    /* R/wavefront_penalties.c:39-64, R/wavefront_components.c:43-56 */
    ws->ncomp = 1; ws->match = 0; ws->x = 1; ws->o1 = 1; ws->e1 = 1; ws->o2 = 1; ws->e2 = 1; ws->scope = 2;
    return 0;
  }
  if (cfg->distance == WFA_DIST_LINEAR) {
    /* R/wavefront_penalties.c:65-94 (pywfa passes gap_extension as the indel penalty, align.pyx:351-355) */
    if (cfg->match > 0 || cfg->mismatch <= 0 || cfg->gap_extension <= 0) return -1;
    ws->ncomp = 1;
    if (cfg->match < 0) {
      ws->match = cfg->match;
      ws->x = 2 * cfg->mismatch - 2 * cfg->match;
      ws->o1 = 2 * cfg->gap_extension - cfg->match;
    } else {
      ws->match = 0; ws->x = cfg->mismatch; ws->o1 = cfg->gap_extension;
    }
    ws->e1 = ws->o2 = ws->e2 = 1;
    ws->scope = MAX2(ws->x, ws->o1) + 1; /* R/wavefront_components.c:57-74 */
    return 0;
  }
  if (cfg->distance != WFA_DIST_AFFINE && cfg->distance != WFA_DIST_AFFINE2P) return -1;
  const int two = (cfg->distance == WFA_DIST_AFFINE2P);
  if (cfg->match > 0 || cfg->mismatch <= 0 || cfg->gap_opening < 0 || cfg->gap_extension <= 0) return -1;
  if (two && (cfg->gap_opening2 < 0 || cfg->gap_extension2 <= 0)) return -1;
  ws->ncomp = two ? 5 : 3;
  if (cfg->match < 0) {
    ws->match = cfg->match;
    ws->x = 2 * cfg->mismatch - 2 * cfg->match;
    ws->o1 = 2 * cfg->gap_opening;
    ws->e1 = 2 * cfg->gap_extension - cfg->match;
    ws->o2 = 2 * cfg->gap_opening2;
    ws->e2 = 2 * cfg->gap_extension2 - cfg->match;
  } else {
    ws->match = 0;
    ws->x = cfg->mismatch;
    ws->o1 = cfg->gap_opening;
    ws->e1 = cfg->gap_extension;
    ws->o2 = cfg->gap_opening2;
    ws->e2 = cfg->gap_extension2;
  }
  /* R/wavefront_components.c:81-124 */
  int scope_indel = ws->o1 + ws->e1;
  if (two && ws->o2 + ws->e2 > scope_indel) scope_indel = ws->o2 + ws->e2;
  ws->scope = MAX2(scope_indel, ws->x) + 1;
  return 0;
}

/*
 * Public entry: same argument list as wfa_hip_align_batch minus the handle.
 * Returns 0; -1 invalid/unsupported configuration; -2 out of memory.
 */
int wfa_oracle_align_batch(const wfa_hip_config_t* cfg, int64_t n, const uint8_t* seqs,
                           const int64_t* p_off, const int32_t* p_len,
                           const int64_t* t_off, const int32_t* t_len,
                           int32_t* score, int32_t* status,
                           uint8_t* cigar_ops, const int64_t* cigar_off,
                           int64_t* cigar_begin, int32_t* cigar_len) {
  oracle_ws_t ws;
  memset(&ws, 0, sizeof(ws));
  if (ws_set_penalties(&ws, cfg)) return -1;
  /* match<0 with free begins (the ends-free re-seeding of R/wavefront_compute.c:124-254): score scope only.  With a backtrace
   * the reference itself fails on ordinary inputs — exit(-1) "I?/D?-Beginning backtrace error" in memory mode high, an endless
   * loop in medium / low (tests/test_oracle_vs_ref.py keeps the reproducers) — so there is nothing to restate. */
  if (cfg->distance >= WFA_DIST_LINEAR && cfg->match < 0 && cfg->span == WFA_SPAN_ENDSFREE &&
      (cfg->pattern_begin_free > 0 || cfg->text_begin_free > 0) && cfg->scope == WFA_SCOPE_FULL) return -1;
  /* BiWFA (R/wavefront_bialign.c): without heuristic, free ends (the reference exit(1)s, R/wavefront_align.c:60-75) or a
   * step limit.  scope=score: wavefront_bialign_compute_score (:662-702) returns what the other memory modes return
   * (pinned by tests/test_oracle_vs_ref.py); scope=full: the breakpoint recursion restated above. */
  int biwfa_full = 0, biwfa_score = 0;
  bi_ctx_t* bcx = NULL;
  if (cfg->memory_mode == WFA_MEM_BIWFA) {
    const int free_ends = cfg->span == WFA_SPAN_ENDSFREE &&
        (cfg->pattern_begin_free | cfg->pattern_end_free | cfg->text_begin_free | cfg->text_end_free) != 0;
    if (free_ends) return -1;
    /* scope=score with a step limit: the limit counts the forward + reverse scores (R/wavefront_bialign.c:475,513), which is
     * not what the ordinary algorithm counts: the top-level breakpoint search is run as it is (:662-702).  With a heuristic
     * likewise: both directions cut their wavefronts off (round 4) */
    if (cfg->scope == WFA_SCOPE_FULL || cfg->max_steps > 0 || cfg->heuristic != WFA_HEUR_NONE) {
      biwfa_full = (cfg->scope == WFA_SCOPE_FULL);
      biwfa_score = !biwfa_full;
      bcx = (bi_ctx_t*)calloc(1, sizeof(bi_ctx_t));
      if (!bcx) return -2;
      if (ws_set_penalties(&bcx->fwd.ws, cfg) || ws_set_penalties(&bcx->rev.ws, cfg) || ws_set_penalties(&bcx->base.ws, cfg)) { free(bcx); return -1; }
      bcx->wc = cfg->wildcard;
      bcx->cfg = cfg;
      bcx->max_steps = (cfg->max_steps <= 0) ? INT_MAX : cfg->max_steps;   /* align.pyx:415-417 -> R/wavefront_bialigner.c:168-174 */
    }
  }
  const int full = (cfg->scope == WFA_SCOPE_FULL);
  if (full && (!cigar_ops || !cigar_off || !cigar_begin || !cigar_len)) return -1;
  int64_t i;
  int rc = 0;
  for (i = 0; i < n; ++i) {
    const int plen = p_len[i], tlen = t_len[i];
    if (cfg->span == WFA_SPAN_ENDSFREE &&
        (cfg->pattern_begin_free > plen || cfg->pattern_end_free > plen ||
         cfg->text_begin_free > tlen || cfg->text_end_free > tlen)) { rc = -1; break; }
    int64_t ob = 0;
    int32_t ol = 0;
    int32_t sc = 0, st = 0;
    if (biwfa_full) {
      if (bi_align_one(bcx, cfg, seqs + p_off[i], plen, seqs + t_off[i], tlen, &sc, &st, cigar_ops + cigar_off[i], &ob, &ol)) { rc = -2; break; }
    } else if (biwfa_score) {
      if (bi_score_one(bcx, seqs + p_off[i], plen, seqs + t_off[i], tlen, &sc, &st)) { rc = -2; break; }
    } else if (align_one(&ws, cfg, seqs + p_off[i], plen, seqs + t_off[i], tlen, &sc, &st,
                         full ? cigar_ops + cigar_off[i] : NULL, &ob, &ol)) { rc = -2; break; }
    score[i] = sc;
    status[i] = st;
    if (cigar_begin) cigar_begin[i] = (full && cigar_off) ? cigar_off[i] + ob : 0;
    if (cigar_len) cigar_len[i] = ol;
  }
  ws_free(&ws);
  if (bcx) { ws_free(&bcx->fwd.ws); ws_free(&bcx->rev.ws); ws_free(&bcx->base.ws); free(bcx); }
  return rc;
}

/*
 * Size-independent property check used by the full-size GPU tests: for every pair the op string
 * must be a valid transcript (M on equal bytes, X on different bytes, I/D consume text/pattern,
 * both sequences fully consumed) and, for end-to-end alignments with match == 0, the gap-affine
 * (or 2-piece) penalty of the transcript must equal -score (cigar_score_gap_affine semantics,
 * R/alignment/cigar.c).  Returns the number of failing pairs; first_bad receives the first index.
 */
int64_t wfa_oracle_check_cigars(const wfa_hip_config_t* cfg, int64_t n, const uint8_t* seqs,
                                const int64_t* p_off, const int32_t* p_len,
                                const int64_t* t_off, const int32_t* t_len,
                                const int32_t* score, const uint8_t* cigar_ops,
                                const int64_t* cigar_begin, const int32_t* cigar_len,
                                int check_score, int64_t* first_bad) {
  const int two = (cfg->distance == WFA_DIST_AFFINE2P);
  int64_t bad = 0, i;
  if (first_bad) *first_bad = -1;
  for (i = 0; i < n; ++i) {
    const uint8_t* P = seqs + p_off[i];
    const uint8_t* T = seqs + t_off[i];
    const uint8_t* ops = cigar_ops + cigar_begin[i];
    const int len = cigar_len[i];
    int v = 0, h = 0, ok = 1, j = 0;
    int64_t pen = 0;
    while (j < len && ok) {
      const uint8_t op = ops[j];
      int run = 1;
      while (j + run < len && ops[j + run] == op) ++run;
      int r;
      switch (op) {
        case 'M':
          for (r = 0; r < run; ++r) { if (v >= p_len[i] || h >= t_len[i] || P[v] != T[h]) { ok = 0; break; } ++v; ++h; }
          break;
        case 'X':
          for (r = 0; r < run; ++r) { if (v >= p_len[i] || h >= t_len[i] || P[v] == T[h]) { ok = 0; break; } ++v; ++h; }
          pen += (int64_t)cfg->mismatch * run;
          break;
        case 'I':
        case 'D': {
          int64_t g = cfg->gap_opening + (int64_t)cfg->gap_extension * run;
          if (two) {
            const int64_t g2 = cfg->gap_opening2 + (int64_t)cfg->gap_extension2 * run;
            if (g2 < g) g = g2;
          }
          pen += g;
          if (op == 'I') h += run; else v += run;
          break;
        }
        default: ok = 0;
      }
      j += run;
    }
    if (ok && (v != p_len[i] || h != t_len[i])) ok = 0;
    if (ok && check_score && pen != -(int64_t)score[i]) ok = 0;
    if (!ok) { if (!bad && first_bad) *first_bad = i; ++bad; }
  }
  return bad;
}
