/*
 * wfa_hip.h — C ABI of the MI355X-native batched wavefront aligner (libwfa_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of kcleal/pywfa: the calls that
 * pywfa's Cython host (pywfa/align.pyx) makes into the vendored WFA2-lib, i.e.
 *
 *   wavefront_aligner_new(&attributes)                       align.pyx:344,419   wfa.h:125-126
 *   wavefront_align(aligner, pattern, plen, text, tlen)      align.pyx:439       wfa.h:199-204
 *   wavefront_align_lambda(...)  (wildcard matching)         align.pyx:441-442   wfa.h:205-210
 *   wavefront_aligner_delete(aligner)                        align.pyx:881-883   wfa.h:129-130
 *   reads of aligner->cigar->{score,operations,begin_offset,end_offset}
 *            aligner->align_status.status                    align.pyx:443,461-467,737-786
 *   writes to aligner->{alignment_form,alignment_scope,heuristic,penalties,...}
 *                                                            align.pyx:469-729
 *
 * Every entry point below is plain C (pointers + sizes, no torch / HIP types in the
 * signatures; a stream is passed as an opaque void*).  The reference aligns ONE pair per
 * call on one CPU thread; the replacement aligns a BATCH of independent pairs per call on
 * one GPU, so the batch forms are additive while the config struct carries exactly the kwargs
 * of WavefrontAligner.__init__ (align.pyx:309-334).  How the pairs map onto the GPU is the
 * library's business and depends on the reads: SIXTY-FOUR alignments per wavefront (a lane
 * each) for short reads — the BASELINE C2 kernel; a measured departure from north_star's
 * "one alignment per workgroup", DESIGN.md §3.0 —, one alignment per wave for long reads
 * under a heuristic, one alignment per workgroup with the M / I / D wavefronts in LDS or in
 * LDS tiles for exact long reads.  Results never depend on the mapping.
 *
 * Error model: the reference calls exit(1) on invalid penalties / ends-free sizes
 * (wavefront_penalties.c:101-112, wavefront_align.c:95-101).  Here every function returns
 * WFA_HIP_OK (0) or a negative WFA_HIP_E* code and wfa_hip_last_error() gives the text.
 * Per-pair results use the reference's own status codes (wfa.h:46-51):
 *   0 completed, 1 partial (heuristically dropped), -100 max steps reached, -200 OOM, -300 unattainable.
 */
#ifndef WFA_HIP_H_
#define WFA_HIP_H_

#ifndef __HIPCC_RTC__   /* (the kernel headers include this file for the status codes, also when compiled by hipRTC) */
#include <stdint.h>
#include <stddef.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* distance_metric_t values of the reference (wavefront_penalties.h:42-48) */
#define WFA_DIST_INDEL     0
#define WFA_DIST_EDIT      1
#define WFA_DIST_LINEAR    2
#define WFA_DIST_AFFINE    3
#define WFA_DIST_AFFINE2P  4

#define WFA_SCOPE_SCORE    0   /* compute_score      (align.pyx:374-375) */
#define WFA_SCOPE_FULL     1   /* compute_alignment  (align.pyx:372-373) */

#define WFA_SPAN_END2END   0   /* alignment_end2end  (align.pyx:396-397) */
#define WFA_SPAN_ENDSFREE  1   /* alignment_endsfree (align.pyx:394-395) */

#define WFA_HEUR_NONE      0   /* wf_heuristic_none        (align.pyx:401-402) */
#define WFA_HEUR_ADAPTIVE  1   /* wf_heuristic_wfadaptive  (align.pyx:403-407) */
#define WFA_HEUR_XDROP     2   /* wf_heuristic_xdrop       (align.pyx:408-411) */

#define WFA_MEM_HIGH       0   /* wavefront_memory_high (explicit wavefront history)  */
#define WFA_MEM_MED        1   /* wavefront_memory_med  (same results as high)        */
#define WFA_MEM_LOW        2   /* wavefront_memory_low  (same results as high)        */
#define WFA_MEM_BIWFA      3   /* wavefront_memory_ultralow (BiWFA, wavefront_bialign.c) — without heuristic / free ends */

/* per-pair status codes, identical to the reference (wfa.h:46-51) */
#define WFA_STATUS_COMPLETED          0
#define WFA_STATUS_PARTIAL            1
#define WFA_STATUS_MAX_STEPS_REACHED  (-100)
#define WFA_STATUS_OOM                (-200)
#define WFA_STATUS_UNATTAINABLE        (-300)  /* BiWFA: no alignment under the configuration (wfa.h:51) */

/* library return codes */
#define WFA_HIP_OK            0
#define WFA_HIP_EINVAL       (-1)   /* invalid configuration / arguments (reference: exit(1)) */
#define WFA_HIP_ENOTSUP      (-2)   /* configuration outside the accelerated path             */
#define WFA_HIP_EDEVICE      (-3)   /* HIP runtime error (no GPU, launch failure, OOM)         */

/*
 * Configuration = kwargs of pywfa.WavefrontAligner.__init__ (align.pyx:309-334), as a POD.
 * Defaults (wfa_hip_config_default) are pywfa's: affine 0/4/6/2 (24/1), scope full,
 * span ends-free with all free ends 0, no heuristic, memory high, max_steps unlimited.
 */
typedef struct wfa_hip_config {
  int32_t distance;                /* WFA_DIST_*                                    */
  int32_t match;                   /* <= 0                                          */
  int32_t mismatch;                /* > 0                                           */
  int32_t gap_opening;             /* >= 0                                          */
  int32_t gap_extension;           /* > 0                                           */
  int32_t gap_opening2;            /* >= 0 (affine2p)                               */
  int32_t gap_extension2;          /* > 0  (affine2p)                               */
  int32_t scope;                   /* WFA_SCOPE_*                                   */
  int32_t span;                    /* WFA_SPAN_*                                    */
  int32_t pattern_begin_free;
  int32_t pattern_end_free;
  int32_t text_begin_free;
  int32_t text_end_free;
  int32_t heuristic;               /* WFA_HEUR_*                                    */
  int32_t min_wavefront_length;    /* adaptive                                      */
  int32_t max_distance_threshold;  /* adaptive                                      */
  int32_t steps_between_cutoffs;   /* adaptive, X-drop                              */
  int32_t xdrop;                   /* X-drop                                        */
  int32_t memory_mode;             /* WFA_MEM_*                                     */
  int32_t max_steps;               /* <= 0: unlimited (align.pyx:415-417)           */
  int32_t wildcard;                /* -1: none; else the byte that matches anything */
  int32_t reserved;                /* must be 0                                     */
} wfa_hip_config_t;

typedef struct wfa_hip_aligner wfa_hip_aligner_t;  /* replaces wavefront_aligner_t*      */
typedef struct wfa_hip_batch   wfa_hip_batch_t;    /* a batch of pairs resident in HBM   */

/* ---- library / device ------------------------------------------------------------------ */

/* ABI version of this header (bumped on any signature change). */
int wfa_hip_abi_version(void);
/* Number of visible HIP devices, or a negative WFA_HIP_E* code. */
int wfa_hip_device_count(void);
/* Text for the last error on this thread when no aligner handle exists (create failed). */
const char* wfa_hip_global_error(void);

/* ---- aligner handle -------------------------------------------------------------------- */

/* Fill *cfg with pywfa's defaults (align.pyx:309-334; wavefront_attributes.c:38-100). */
int wfa_hip_config_default(wfa_hip_config_t* cfg);
/* Validate like wavefront_penalties_set_* (wavefront_penalties.c:95-173) but return
 * WFA_HIP_EINVAL instead of exit(1). err (nullable) receives a message of at most errlen. */
int wfa_hip_config_validate(const wfa_hip_config_t* cfg, char* err, size_t errlen);

/* Replaces wavefront_aligner_new (wfa.h:125-126). device = HIP device ordinal.
 * Returns NULL on error (see wfa_hip_global_error). */
wfa_hip_aligner_t* wfa_hip_create(const wfa_hip_config_t* cfg, int device);
/* Replaces wavefront_aligner_delete (wfa.h:129-130). */
void wfa_hip_destroy(wfa_hip_aligner_t* aligner);
/* Replaces pywfa's property setters that poke C fields after construction
 * (align.pyx:469-729): swap in a new validated configuration. */
int wfa_hip_set_config(wfa_hip_aligner_t* aligner, const wfa_hip_config_t* cfg);
int wfa_hip_get_config(const wfa_hip_aligner_t* aligner, wfa_hip_config_t* cfg);
const char* wfa_hip_last_error(const wfa_hip_aligner_t* aligner);

/* ---- host-buffer batch alignment (the drop-in for N x wavefront_align) ------------------ */

/*
 * Align n independent (pattern, text) pairs; synchronous.
 *   seqs            ASCII bytes, compared raw like the reference (wavefront_sequences.c:250);
 *                   pair i uses seqs[p_off[i] .. +p_len[i]) and seqs[t_off[i] .. +t_len[i])
 *   score[i]        what aligner->cigar->score holds after wavefront_align (align.pyx:443)
 *   status[i]       what aligner->align_status.status holds (align.pyx:461-463)
 *   cigar_ops       (nullable unless scope=full) caller buffer; pair i owns the region
 *                   [cigar_off[i], cigar_off[i+1]) which must hold >= p_len[i]+t_len[i] bytes
 *   cigar_off       n+1 region starts (host-computed prefix sums)
 *   cigar_begin[i], cigar_len[i]
 *                   the op string cigar->operations[begin_offset:end_offset) of pair i is
 *                   cigar_ops[cigar_begin[i] .. +cigar_len[i])  (chars M X I D)
 * All buffers are borrowed for the call; outputs are caller-owned.
 * Inside: up to 4 096 short pairs take one launch on a pinned block (single calls of a pywfa-style loop: ~30 us); batches of
 * >= 256 k pairs are packed to 2 bits per base by host threads on their way into a pinned upload ring; everything else is
 * uploaded as it is and packed on the device.  The results are the same whichever way.
 */
int wfa_hip_align_batch(wfa_hip_aligner_t* aligner, int64_t n,
                        const uint8_t* seqs,
                        const int64_t* p_off, const int32_t* p_len,
                        const int64_t* t_off, const int32_t* t_len,
                        int32_t* score, int32_t* status,
                        uint8_t* cigar_ops, const int64_t* cigar_off,
                        int64_t* cigar_begin, int32_t* cigar_len);

/*
 * One pair per call — pywfa's own usage pattern: wavefront_align(text) against the cached pattern (align.pyx:421-443, one
 * wavefront_align / wavefront_align_lambda per call, wfa.h:199-210).  The result of wfa_hip_align_batch with n = 1 without its
 * arrays: `pattern` / `text` are the two ASCII sequences; `cigar_ops` (NULL for scope = score) receives plen + tlen bytes of which
 * [*cigar_begin, *cigar_begin + *cigar_len) are the alignment's ops.  Round 6: calls in a row are served without a kernel launch — a
 * one-wave kernel stays on the device and takes the pairs from a mailbox in pinned host memory (it leaves by itself after 2 ms without
 * a call, before any batch of this aligner, and when the aligner is destroyed; gap-affine shapes of the library, reads of up to 1 000
 * bases; anything else takes the single-launch path: the pair packed into a pinned block, one wave, a completion flag polled by the
 * host).  About 9 us per 150 bp call from C score-only, 16.5 us with the op string, against 1-2 us for the reference on a host core —
 * the library is batch-oriented, and a loop of single calls, while exact, is what wfa_hip_align_batch with many pairs replaces.
 * WFA_HIP_MAILBOX=0: a launch per call (13 / 21 us).
 */
int wfa_hip_align_pair(wfa_hip_aligner_t* aligner, const uint8_t* pattern, int32_t plen, const uint8_t* text, int32_t tlen,
                       int32_t* score, int32_t* status, uint8_t* cigar_ops, int64_t* cigar_begin, int32_t* cigar_len);

/*
 * The same call for a caller that already holds 2-bit reads (cf. wavefront_align_packed2bits, wfa.h:211-216,
 * wavefront_sequences.h:115): `packed` holds every sequence in the reference's packed form (wavefront_sequences.c:102-139:
 * four bases per byte, base j of a byte in bits 2j .. 2j+1, 'A' 0 / 'C' 1 / 'G' 2 / 'T' 3), a sequence of len bases being
 * (len + 3) / 4 bytes starting at its BYTE offset p_off[i] / t_off[i]; p_len / t_len stay in bases.  The results are those
 * of wfa_hip_align_batch on the decoded ASCII sequences (that is what the tests pin it against: the reference's own entry
 * reads only (len + 7) / 8 bytes per sequence, wavefront_sequences.c:112, and aligns uninitialised buffer bytes behind them).
 * A quarter of the bytes are read on the host; large batches are re-based to whole words by the upload workers on their way into the
 * pinned ring (round 6: the same bytes cross PCIe as for ASCII input).  A wildcard letter cannot be expressed: WFA_HIP_ENOTSUP.
 */
int wfa_hip_align_batch_packed2bits(wfa_hip_aligner_t* aligner, int64_t n,
                                    const uint8_t* packed,
                                    const int64_t* p_off, const int32_t* p_len,
                                    const int64_t* t_off, const int32_t* t_len,
                                    int32_t* score, int32_t* status,
                                    uint8_t* cigar_ops, const int64_t* cigar_off,
                                    int64_t* cigar_begin, int32_t* cigar_len);

/* ---- host helper: the text of cigar_print_pretty ------------------------------------------------- */

/*
 * What cigar_print_pretty prints (cigar.h:180-186, cigar.c:778-863; called by align.pyx:445-459) for the op string
 * ops[0 .. ops_len) (chars M X I D, i.e. cigar->operations[begin_offset .. end_offset)) of `pattern` against `text`: the
 * ALIGNMENT (runs incl. M), ETRACE (runs without M) and CIGAR (SAM style, X folded into M) lines and the three rows
 * PATTERN / marks / TEXT.  Written NUL-terminated into out[0 .. cap) (truncated if cap is too small, like snprintf);
 * returns the length of the whole text without the NUL, or WFA_HIP_EINVAL.  Host only, needs no GPU.
 */
int64_t wfa_hip_cigar_sprint_pretty(const uint8_t* ops, int64_t ops_len,
                                    const uint8_t* pattern, int32_t plen,
                                    const uint8_t* text, int32_t tlen,
                                    char* out, int64_t cap);

/* ---- host helper: the 2-bit packing the large-batch upload uses ----------------------------------- */

/*
 * Packs `len` ASCII bases into (len + 15) / 16 words: word w holds bases 16 w .. 16 w + 15, base j in bits 2 j .. 2 j + 1,
 * code (c >> 1) & 3 ('A' 0, 'C' 1, 'T' 2, 'G' 3), zero beyond the end — the layout the kernels read.  Host only, needs no
 * GPU.  wfa_hip_align_batch / wfa_hip_batch_create run it on several host threads for batches of >= 256 k pairs, so
 * that 2 bits per base cross PCIe instead of 8 (the reference has no counterpart: it reads the caller's bytes in place,
 * wavefront_sequences.c:153-250).  form: -1 = the best the CPU has; 0 plain C, 1 AVX2, 2 AVX-512BW (for tests; a form the
 * CPU lacks falls back to plain C).  Returns 1 if some byte is not one of ACGT (such a pair is aligned on its bytes),
 * 0 if none, WFA_HIP_EINVAL on bad arguments.
 */
int wfa_hip_pack_2bit(const uint8_t* seq, int32_t len, uint32_t* words, int form);

/*
 * The number of blob bytes a batch description reaches: max over pairs of p_off + p_len and t_off + t_len (host only,
 * several threads), or -1 if an offset or a length is negative or an argument is missing.  For bindings: the align /
 * create calls take the blob by pointer only, so a binding that knows the blob's size checks it against this first
 * (pywfa_amd/_native.py does; pywfa itself passes one str per call, align.pyx:432-437).
 */
int64_t wfa_hip_batch_extent(int64_t n, const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len);
/* The same for a batch of 2-bit reads (wfa_hip_align_batch_packed2bits): a sequence of len bases reaches (len + 3) / 4 bytes. */
int64_t wfa_hip_batch_extent_packed2bits(int64_t n, const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len);

/* ---- several devices of one node ------------------------------------------------ */

/*
 * Pairs are independent, so a batch shards over the GPUs of a node with no exchange step: contiguous shards
 * balanced by sum(p_len + t_len), one host thread + one aligner + one stream per device, results written into
 * disjoint slices of the caller's arrays; no collective.  (The reference has no counterpart: it aligns one pair per
 * call on one CPU thread.)
 */
typedef struct wfa_hip_multi wfa_hip_multi_t;

/* The shard planner alone (host only, needs no GPU): shard_begin[nshards + 1], shard s = pairs
 * [shard_begin[s], shard_begin[s + 1]).  Returns WFA_HIP_OK or WFA_HIP_EINVAL. */
int wfa_hip_plan_shards(int64_t n, const int32_t* p_len, const int32_t* t_len, int nshards, int64_t* shard_begin);

/* The host side of several devices (host only, needs no GPU): the threads ONE device's upload pipeline takes — 2-bit packing into its
 * pinned ring, and plain copies — when `sharers` aligners or processes feed GPUs from a host of `hw_threads` logical CPUs.  The
 * devices of a wfa_hip_multi_t count themselves; a one-process-per-GPU job says so through LOCAL_WORLD_SIZE (torch.distributed.run
 * exports it) or WFA_HIP_HOST_SHARE.  Returns WFA_HIP_OK or WFA_HIP_EINVAL. */
int wfa_hip_plan_host_threads(int sharers, int hw_threads, int* pack_threads, int* copy_threads);

/* What the upload pipeline of this aligner knows about the host, and what its last pipelined upload (a batch of >= 256 k pairs) did
 * (diagnostics for the PCIe-inclusive rate; no counterpart in the reference, whose aligner never leaves the host).  info[0..7]:
 * NUMA node of the device's PCIe slot (-1 unknown), CPUs of that node this process may use (0: no binding possible), binding mode
 * (WFA_HIP_NUMA: 0 never, 1 always, 2 only when the caller's input lives on the device's node), node of the caller's pages in the last
 * upload (-1 unknown), whether that upload's spawned workers were bound (the caller's own thread never is), packing threads, copy
 * threads, CPUs of the process.  n >= 8.  Returns WFA_HIP_OK or WFA_HIP_EINVAL. */
int wfa_hip_upload_info(const wfa_hip_aligner_t* aligner, int32_t* info, int n);

/* One aligner per entry of devices[] (an ordinal may repeat: several host threads then feed that device).
 * Returns NULL on error (see wfa_hip_global_error). */
wfa_hip_multi_t* wfa_hip_multi_create(const wfa_hip_config_t* cfg, const int* devices, int ndevices);
void wfa_hip_multi_destroy(wfa_hip_multi_t* multi);
int wfa_hip_multi_set_config(wfa_hip_multi_t* multi, const wfa_hip_config_t* cfg);
const char* wfa_hip_multi_last_error(const wfa_hip_multi_t* multi);
/* Same arguments and results as wfa_hip_align_batch; synchronous. */
int wfa_hip_multi_align_batch(wfa_hip_multi_t* multi, int64_t n,
                              const uint8_t* seqs,
                              const int64_t* p_off, const int32_t* p_len,
                              const int64_t* t_off, const int32_t* t_len,
                              int32_t* score, int32_t* status,
                              uint8_t* cigar_ops, const int64_t* cigar_off,
                              int64_t* cigar_begin, int32_t* cigar_len);

/* ---- HBM-resident batches (what bench.py times; inputs resident before the clock starts) -- */

/* Upload n pairs (same input arrays as above) as 2-bit codes (packed by host threads on their way into the pinned upload
 * ring for batches of >= 256 k pairs, by a device kernel otherwise).  The returned batch keeps sequences, per-pair metadata
 * and result arrays in HBM.  For short-read batches of >= 64 k pairs the call also runs the pilot that picks the first
 * stage of the cascade (up to three small launches on 8192 pairs sampled across the batch, each waited for).  The input
 * arrays are not read after the call returns; uploads still in flight then are ordered before any later run by the library
 * (whatever stream the run is given). */
wfa_hip_batch_t* wfa_hip_batch_create(wfa_hip_aligner_t* aligner, int64_t n,
                                      const uint8_t* seqs,
                                      const int64_t* p_off, const int32_t* p_len,
                                      const int64_t* t_off, const int32_t* t_len);
/* The same for 2-bit reads in the reference's packed form (see wfa_hip_align_batch_packed2bits). */
wfa_hip_batch_t* wfa_hip_batch_create_packed2bits(wfa_hip_aligner_t* aligner, int64_t n,
                                                  const uint8_t* packed,
                                                  const int64_t* p_off, const int32_t* p_len,
                                                  const int64_t* t_off, const int32_t* t_len);
void wfa_hip_batch_destroy(wfa_hip_batch_t* batch);
/* Enqueue the alignment kernels for the whole batch on `stream` (hipStream_t passed as
 * void*, NULL = the library's own stream) and return without waiting: in the steady state the call only enqueues (kernels,
 * memsets, event waits).  Exceptions, all one-off: the FIRST run of an aligner and any run that has to grow its workspace
 * allocate device memory (hipFree / hipMalloc synchronise the device, and may drain the aligner's block pool), and the first
 * run under penalties the library has no built-in kernels for compiles them (hipRTC, about a second per kernel; cached on
 * disk under ~/.cache/pywfa_amd or $WFA_HIP_RTC_CACHE). */
int wfa_hip_batch_run(wfa_hip_batch_t* batch, void* stream);
/* Wait for the last run of this batch. */
int wfa_hip_batch_sync(wfa_hip_batch_t* batch);
/* Copy results of the last run to host arrays (cigar_* nullable for scope=score). */
int wfa_hip_batch_results(wfa_hip_batch_t* batch, int32_t* score, int32_t* status,
                          uint8_t* cigar_ops, const int64_t* cigar_off,
                          int64_t* cigar_begin, int32_t* cigar_len);
/* Mean HIP-event time (ms) per run of the alignment kernels, over the runs enqueued since the
 * previous sync (events are recorded on the stream the kernels are launched on), and the number of
 * pairs one run hands to the dominant kernel (for bench.py's roofline line). */
int wfa_hip_batch_last_kernel_ms(wfa_hip_batch_t* batch, float* ms, int64_t* pairs);
/* Algorithmic HBM bytes of one run: 2-bit packed sequence bytes + 8 result bytes per pair
 * (+ CIGAR op bytes for scope=full), SURVEY.md §8(d). */
int64_t wfa_hip_batch_algorithmic_bytes(const wfa_hip_batch_t* batch);
/* Pairs the fast (register/LDS) kernel handed to the general kernel in the last run. */
int64_t wfa_hip_batch_fallback_pairs(const wfa_hip_batch_t* batch);

/* ---- result surface on the device ------------------------------------------ */

/*
 * What pywfa derives per pair in Python from cigar->operations after each call — the run-length
 * encoded `cigartuples` (align.pyx:759-786; op codes M=0 I=1 D=2 X=8) and the `locations`
 * (pattern_start, pattern_end, text_start, text_end; align.pyx:788-833) — for the whole batch of the
 * last scope=full run, computed on the GPU.
 *   step 1  wfa_hip_batch_rle_counts: run_count[n] and locations[4n]; returns the total number of runs
 *           (or a negative WFA_HIP_E* code)
 *   step 2  wfa_hip_batch_rle_runs: run_code[total], run_len[total] in pair order
 *           (pair i owns runs [sum(run_count[:i]), +run_count[i]))
 */
int64_t wfa_hip_batch_rle_counts(wfa_hip_batch_t* batch, int32_t* run_count, int32_t* locations);
int wfa_hip_batch_rle_runs(wfa_hip_batch_t* batch, uint8_t* run_code, int32_t* run_len);

#ifdef __cplusplus
}
#endif
#endif /* WFA_HIP_H_ */
