/*
 * oracle/ref_harness.c — TEST INFRASTRUCTURE, not product code.
 *
 * A thin batch loop over the REAL reference library (WFA2-lib v2.3 as vendored by pywfa),
 * compiled by oracle/Makefile straight from the sources where they lie under
 * /root/reference/pywfa/WFA2_lib (nothing is copied into this repository) into
 * oracle/_ref/libwfa_ref.so.  It builds the aligner attributes from a wfa_hip_config_t
 * exactly the way pywfa's Cython host does (/root/reference/pywfa/align.pyx:344-419) and
 * calls wavefront_align() once per pair (align.pyx:439), reading back what pywfa reads:
 * cigar->score (align.pyx:443), align_status.status (:461-463) and
 * cigar->operations[begin_offset:end_offset) (:737-786).
 *
 * Used only by tests/, tools/make_golden.py, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg (kind "reference").
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#include "wavefront/wavefront_align.h"
#include "wfa_hip.h"

typedef struct {
  const char* pattern;
  const char* text;
  char wildcard;
} ref_wildcard_args_t;

/* same predicate as align.pyx:302-304 */
static int ref_wildcard_match(int v, int h, void* argsptr) {
  const ref_wildcard_args_t* a = (const ref_wildcard_args_t*)argsptr;
  return a->pattern[v] == a->wildcard || a->text[h] == a->wildcard || a->pattern[v] == a->text[h];
}

static wavefront_aligner_t* ref_new_aligner(const wfa_hip_config_t* cfg) {
  wavefront_aligner_attr_t attributes = wavefront_aligner_attr_default;
  switch (cfg->distance) {
    case WFA_DIST_INDEL: attributes.distance_metric = indel; break;
    case WFA_DIST_EDIT: attributes.distance_metric = edit; break;
    case WFA_DIST_LINEAR:
      attributes.distance_metric = gap_linear;
      attributes.linear_penalties.match = cfg->match;
      attributes.linear_penalties.mismatch = cfg->mismatch;
      attributes.linear_penalties.indel = cfg->gap_extension;
      break;
    case WFA_DIST_AFFINE:
      attributes.distance_metric = gap_affine;
      attributes.affine_penalties.match = cfg->match;
      attributes.affine_penalties.mismatch = cfg->mismatch;
      attributes.affine_penalties.gap_opening = cfg->gap_opening;
      attributes.affine_penalties.gap_extension = cfg->gap_extension;
      break;
    case WFA_DIST_AFFINE2P:
      attributes.distance_metric = gap_affine_2p;
      attributes.affine2p_penalties.match = cfg->match;
      attributes.affine2p_penalties.mismatch = cfg->mismatch;
      attributes.affine2p_penalties.gap_opening1 = cfg->gap_opening;
      attributes.affine2p_penalties.gap_extension1 = cfg->gap_extension;
      attributes.affine2p_penalties.gap_opening2 = cfg->gap_opening2;
      attributes.affine2p_penalties.gap_extension2 = cfg->gap_extension2;
      break;
    default: return NULL;
  }
  attributes.alignment_scope = (cfg->scope == WFA_SCOPE_FULL) ? compute_alignment : compute_score;
  switch (cfg->memory_mode) {
    case WFA_MEM_HIGH: attributes.memory_mode = wavefront_memory_high; break;
    case WFA_MEM_MED: attributes.memory_mode = wavefront_memory_med; break;
    case WFA_MEM_LOW: attributes.memory_mode = wavefront_memory_low; break;
    case WFA_MEM_BIWFA: attributes.memory_mode = wavefront_memory_ultralow; break;
    default: return NULL;
  }
  attributes.alignment_form.pattern_begin_free = cfg->pattern_begin_free;
  attributes.alignment_form.pattern_end_free = cfg->pattern_end_free;
  attributes.alignment_form.text_begin_free = cfg->text_begin_free;
  attributes.alignment_form.text_end_free = cfg->text_end_free;
  attributes.alignment_form.span =
      (cfg->span == WFA_SPAN_ENDSFREE) ? alignment_endsfree : alignment_end2end;
  if (cfg->heuristic == WFA_HEUR_NONE) {
    attributes.heuristic.strategy = wf_heuristic_none;
  } else if (cfg->heuristic == WFA_HEUR_ADAPTIVE) {
    attributes.heuristic.strategy = wf_heuristic_wfadaptive;
    attributes.heuristic.min_wavefront_length = cfg->min_wavefront_length;
    attributes.heuristic.max_distance_threshold = cfg->max_distance_threshold;
    attributes.heuristic.steps_between_cutoffs = cfg->steps_between_cutoffs;
  } else if (cfg->heuristic == WFA_HEUR_XDROP) {
    attributes.heuristic.strategy = wf_heuristic_xdrop;
    attributes.heuristic.xdrop = cfg->xdrop;
    attributes.heuristic.steps_between_cutoffs = cfg->steps_between_cutoffs;
  } else {
    return NULL;
  }
  attributes.system.max_alignment_steps = (cfg->max_steps <= 0) ? INT_MAX : cfg->max_steps;
  return wavefront_aligner_new(&attributes);
}

/*
 * Align n pairs one after the other on the calling thread with ONE aligner object
 * (the way a pywfa user re-uses one WavefrontAligner).  Returns 0, or -1 on a bad config.
 * cigar_* are nullable (pass NULL for timing runs / scope=score).
 */
int ref_align_batch(const wfa_hip_config_t* cfg, int64_t n, const uint8_t* seqs,
                    const int64_t* p_off, const int32_t* p_len,
                    const int64_t* t_off, const int32_t* t_len,
                    int32_t* score, int32_t* status,
                    uint8_t* cigar_ops, const int64_t* cigar_off,
                    int64_t* cigar_begin, int32_t* cigar_len) {
  wavefront_aligner_t* const aligner = ref_new_aligner(cfg);
  if (aligner == NULL) return -1;
  int64_t i;
  for (i = 0; i < n; ++i) {
    const char* const pattern = (const char*)(seqs + p_off[i]);
    const char* const text = (const char*)(seqs + t_off[i]);
    if (cfg->wildcard >= 0) {
      ref_wildcard_args_t args = {pattern, text, (char)cfg->wildcard};
      wavefront_align_lambda(aligner, ref_wildcard_match, &args, p_len[i], t_len[i]);
    } else {
      wavefront_align(aligner, pattern, p_len[i], text, t_len[i]);
    }
    if (score) score[i] = aligner->cigar->score;
    if (status) status[i] = aligner->align_status.status;
    if (cigar_len) {
      const cigar_t* const cigar = aligner->cigar;
      int len = cigar->end_offset - cigar->begin_offset;
      if (len < 0) len = 0;
      cigar_len[i] = len;
      if (cigar_begin) cigar_begin[i] = cigar_off ? cigar_off[i] : 0;
      if (cigar_ops && cigar_off && len > 0) {
        memcpy(cigar_ops + cigar_off[i], cigar->operations + cigar->begin_offset, (size_t)len);
      }
    }
  }
  wavefront_aligner_delete(aligner);
  return 0;
}

/* The same loop (no wildcard) on `nthreads` host threads (contiguous slices of the batch, one aligner object per thread):
 * the all-core CPU figure bench.py prints beside the single-thread baseline. */
int ref_align_batch_mt(const wfa_hip_config_t* cfg, int nthreads, int repeat, int64_t n, const uint8_t* seqs,
                       const int64_t* p_off, const int32_t* p_len,
                       const int64_t* t_off, const int32_t* t_len,
                       int32_t* score, int32_t* status) {
  int rc = 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
  for (int t = 0; t < nthreads; ++t) {
    const int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
    if (hi > lo) {
      /* one aligner per thread, its slice walked `repeat` times (amortises the aligner set-up in timing runs) */
      wavefront_aligner_t* const aligner = ref_new_aligner(cfg);
      if (aligner == NULL) { rc = -1; continue; }
      for (int rep = 0; rep < (repeat < 1 ? 1 : repeat); ++rep) {
        for (int64_t i = lo; i < hi; ++i) {
          wavefront_align(aligner, (const char*)(seqs + p_off[i]), p_len[i], (const char*)(seqs + t_off[i]), t_len[i]);
          score[i] = aligner->cigar->score;
          status[i] = aligner->align_status.status;
        }
      }
      wavefront_aligner_delete(aligner);
    }
  }
  return rc;
}

/* ref_align_batch on `nthreads` host threads, op strings included: the pairs are handed out one at a time (long reads differ
 * a lot in cost), one aligner object per thread, created on the thread's first pair.  For the parity checks of the expensive
 * configurations (exact gap-affine-2p at 10 kb: ~0.1 s and ~0.4 GB per pair on one core). */
int ref_align_batch_mt_full(const wfa_hip_config_t* cfg, int nthreads, int64_t n, const uint8_t* seqs,
                            const int64_t* p_off, const int32_t* p_len,
                            const int64_t* t_off, const int32_t* t_len,
                            int32_t* score, int32_t* status,
                            uint8_t* cigar_ops, const int64_t* cigar_off,
                            int64_t* cigar_begin, int32_t* cigar_len) {
  int rc = 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    wavefront_aligner_t* aligner = NULL;
#pragma omp for schedule(dynamic, 1)
    for (int64_t i = 0; i < n; ++i) {
      if (aligner == NULL) aligner = ref_new_aligner(cfg);
      if (aligner == NULL) { rc = -1; continue; }
      wavefront_align(aligner, (const char*)(seqs + p_off[i]), p_len[i], (const char*)(seqs + t_off[i]), t_len[i]);
      score[i] = aligner->cigar->score;
      status[i] = aligner->align_status.status;
      if (cigar_len) {
        const cigar_t* const cigar = aligner->cigar;
        int len = cigar->end_offset - cigar->begin_offset;
        if (len < 0) len = 0;
        cigar_len[i] = len;
        if (cigar_begin) cigar_begin[i] = cigar_off ? cigar_off[i] : 0;
        if (cigar_ops && cigar_off && len > 0) memcpy(cigar_ops + cigar_off[i], cigar->operations + cigar->begin_offset, (size_t)len);
      }
    }
    if (aligner != NULL) wavefront_aligner_delete(aligner);
  }
  return rc;
}

const char* ref_version(void) { return "WFA2-lib v2.3 (pywfa 0.5.1 vendored copy)"; }
