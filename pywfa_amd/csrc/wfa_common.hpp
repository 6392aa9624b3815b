// wfa_common.hpp — types shared by the device kernels and the host side of libwfa_hip.so.
//
// Vocabulary follows the reference's domain (WFA2-lib, /root/reference/pywfa/WFA2_lib/wavefront):
//   diagonal k = h - v, offset = h (text position), v = offset - k (wavefront_offset.h:50-57);
//   a wavefront = {lo, hi, offsets[k]} per score and component M / I1 / D1 / I2 / D2
//   (wavefront.h:56-77); NULL offset = INT32_MIN/2 (wavefront_offset.h:44).
#pragma once
#include "wfa_rtc_compat.hpp"

#define WFA_OFFSET_NULL (-1073741824)  // INT32_MIN/2

// per-pair internal status written by kernels (never returned to callers)
#define WFA_INTERNAL_FALLBACK  (-1000)  // fast kernel: pair does not fit its window -> general kernel
#define WFA_INTERNAL_OVERFLOW  (-1001)  // general kernel: wavefront arena too small -> retry with a larger one

// Per-pair metadata resident in HBM (16 B): word offsets of the 2-bit packed sequences and lengths.
struct WfaPairMeta {
  uint32_t p_woff;  // first u32 word of the packed pattern
  uint32_t t_woff;  // first u32 word of the packed text
  int32_t plen;
  int32_t tlen;
};

// Penalties after wavefront_penalties_set_affine/affine2p (wavefront_penalties.c:95-173) and the
// rest of the per-aligner configuration, in the form the kernels consume.
struct WfaDevConfig {
  int32_t metric; // WFA_DIST_* (0 indel, 1 edit, 2 linear, 3 affine, 4 affine2p)
  int32_t match;  // <= 0 (original match score)
  int32_t x, o1, e1, o2, e2;  // adjusted penalties
  int32_t scope;  // max_score_scope (wavefront_components.c:81-124)
  int32_t endsfree;
  int32_t pbf, pef, tbf, tef;  // pattern/text begin/end free
  int32_t heuristic;           // WFA_HEUR_*
  int32_t min_wf_len, max_dist_thr, steps_between, xdrop;
  int32_t max_steps;           // INT32_MAX = unlimited
  int32_t wildcard;            // -1 none
  int32_t score_mode;          // how a completed pair's score is reported from the kernels' -s (csrc/wfa_hip.hip derive_dev_config):
                               // 0 as it is; 1 (sw_match (plen + tlen) - s) / 2 (match < 0, R/wavefront_penalties.h:73); 2 +s (indel / edit)
  int32_t sw_match;            // -match of the original configuration (score_mode 1)
  int32_t rtc;                 // 1: penalty shapes without an instantiation are compiled at run time (csrc/wfa_rtc.cpp)
  int32_t biwfa_top;           // 1: the general kernel stands in for the top-level base case of BiWFA (reads of <= 100 bases whose
                               // score outgrows the BiWFA kernel's base-case history): a completed pair keeps the unset score
                               // (SURVEY Q6), every other ending is "unattainable" (R/wavefront_bialign.c:182-187,725-729)
};

// Arguments of the alignment kernels (passed by value).
struct WfaKernelArgs {
  // sequences
  const uint32_t* words;    // 2-bit packed, 16 bases per u32, base i of a sequence in bits [2i,2i+2)
  const uint8_t* bytes;     // ASCII blob (8-bit path)
  const WfaPairMeta* meta;
  const int64_t* p_boff;    // byte offsets (8-bit path)
  const int64_t* t_boff;
  // work list: pair ids to process (nullptr = identity); count read from *nwork_dev when non-null
  const uint32_t* worklist;
  const uint32_t* nwork_dev;
  uint32_t nwork;
  // results
  int32_t* score;
  int32_t* status;
  uint8_t* cigar_ops;
  const int64_t* cigar_off;
  int64_t* cigar_begin;
  int32_t* cigar_len;
  // per-workgroup wavefront workspace (general kernel)
  int32_t* ws;
  int64_t ws_stride;  // int32 elements per workgroup
  // fallback list produced by the fast kernel
  uint32_t* fb_list;
  uint32_t* fb_count;
  WfaDevConfig cfg;
};
