// wfa_hip.hip — host side of libwfa_hip.so: the C ABI declared in include/wfa_hip.h.
//
// It replaces, for batches of independent pairs, what pywfa's Cython host does per pair through
// WFA2-lib's C API (align.pyx:344-443,461-467,737-786): build the aligner from the kwargs, run
// wavefront_align, read status / score / CIGAR ops.  Plain HIP runtime only (no torch types).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <string>
#include <map>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <memory>
#include <atomic>
#include <mutex>
#include <functional>
#include <ctype.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>

#include "wfa_hip.h"
#include "wfa_common.hpp"
#include "wfa_pack.hpp"
#include "wfa_general.hpp"
#include "wfa_wide.hpp"
#include "wfa_tile.hpp"
#include "wfa_rtc.hpp"
#include "wfa_seg.hpp"
#include "wfa_lane.hpp"
#include "wfa_band.hpp"
#include "wfa_biwfa.hpp"
#include "wfa_bilevel.hpp"
#include "wfa_rle.hpp"

#define WFA_HIP_ABI_VERSION 3

static thread_local std::string g_error;

// Development knobs, read from the environment ONCE per aligner in wfa_hip_create: the hot entry points
// never call getenv.
#define WFA_COUNTER_WORDS 256  // counters of a batch (wfa_hip_batch::d_counters)
#define WFA_KNOBS(F)                                                                                              \
  F(ARENA_KB) F(BAND_DEBUG) F(BAND_SLIM) F(BAND_LDS_MAX) F(BAND_NO_WIN) F(BAND_SPLIT_MIN) F(NO_TINY_INLINE) F(BILEVEL) F(BILEVEL_WIDE_LEVELS) F(BILEVEL_PER_CU) F(BILEVEL_I32) F(BILEVEL_QCAP) F(BILEVEL_LEVELS) F(BILEVEL_LDS) F(BILEVEL_NO_1024) F(BILEVEL_HUGE_MIN) F(BILEVEL_LDS_W) F(BILEVEL_NO_SEQL) F(LANE_DYN) F(LANE_DYN_WAVES) F(BAND_LEFTOVER_WAVES_PER_CU) F(BAND_NCH) F(BAND_NO_LDS) F(BAND_NO_SPLIT) F(BAND_PB) F(PIPE_TAIL) F(LEN16) F(MAILBOX) F(MAILBOX_IDLE_US) F(TILE32)     \
  F(BAND_RECORDS) F(BAND_SPLIT_ROUNDS) F(BAND_WAVES_PER_CU) F(NO_BAND) F(NO_FAST) F(NO_SEGFULL) F(SEGFULL_PAIRS)     \
  F(SEGFULL_STAGES) F(STAGE_TIMING) F(LANE_HEUR32) F(THREADS) F(TINY_BATCH) F(WAVES_PER_CU) F(FAST_WAVES_PER_CU) F(TIMING)          \
  F(LANE_FULL) F(LANE_FULL_SPLIT) F(LANE_HEUR) F(SEG_HEUR) F(LANE_LDS_PAD_KB) F(LANE_MIN_PAIRS) F(PIPE_CHUNK) F(PIPE_THREADS) F(PACK_THREADS) F(NO_TINY_BAND) F(NO_TINY_POLL) F(UP_STREAMS) F(NO_DUAL) F(NO_WIDE) F(WIDE_ADAPT) F(WIDE_GROWS) F(WIDE_LDS_KB) F(WIDE_THREADS) F(TILE) F(TILE_T) F(TILE_WT) F(TILE_THREADS) F(TILE_PER_CU) F(NO_PIPE) F(HOST_PACK) F(GENERAL_PB) F(LANE_WAVES_PER_CU) F(LANE_REFILL_MIN) F(LANE_DEBUG) F(NO_TINY) F(PILOT_PCT) F(WIDE_ADAPT_LDS)
enum WfaKnob {
#define WFA_KNOB_ENUM(n) K_##n,
  WFA_KNOBS(WFA_KNOB_ENUM)
#undef WFA_KNOB_ENUM
  K_COUNT
};
struct WfaKnobs {
  int value[K_COUNT];
  bool set[K_COUNT];
  std::string fast_stages;  // WFA_HIP_FAST_STAGES (digits)
  void load() {
    static const char* const names[K_COUNT] = {
#define WFA_KNOB_NAME(n) "WFA_HIP_" #n,
        WFA_KNOBS(WFA_KNOB_NAME)
#undef WFA_KNOB_NAME
    };
    for (int i = 0; i < K_COUNT; ++i) {
      const char* v = getenv(names[i]);
      set[i] = v && *v;
      value[i] = set[i] ? atoi(v) : 0;
    }
    const char* fs = getenv("WFA_HIP_FAST_STAGES");
    fast_stages = (fs && *fs) ? fs : "";
  }
};

struct wfa_hip_aligner {
  int numa_state = 0;     // 0 not looked up, 1 the device's node and its CPUs are known (`numa_cpus`), 2 no binding ever (no NUMA information, one node, too few CPUs, WFA_HIP_NUMA=0)
  int numa_node = -1;     // NUMA node of the device's PCIe slot
  cpu_set_t numa_cpus;    // that node's CPUs, as far as this process may run on them
  int numa_mode = 0;      // WFA_HIP_NUMA: 0 never bind, 1 always bind the spawned upload workers to the device's node, 2 only when the caller's input lives there
  cpu_set_t proc_cpus;    // the process's affinity mask when the aligner was created (before anything here bound a thread)
  bool proc_cpus_valid = false;
  int last_src_node = -1, last_bound = 0;   // the last pipelined upload: node of the caller's pages (-1 unknown), workers bound or not
  int host_share = 1;     // aligners / processes feeding GPUs from this host (thread plan of the upload pipeline)
  std::string rtc_note;   // why the run-time kernels were switched off (wfa_hip_batch_run), empty otherwise
  int device = 0;
  wfa_hip_config_t cfg;
  WfaDevConfig dcfg;
  int ncomp = 3;
  WfaDevConfig gcfg;      // what the general kernel runs (= dcfg unless dcfg.lin)
  int gncomp = 3;
  WfaKnobs knobs;
  hipStream_t stream = nullptr;
  std::vector<uint8_t> pair_blob;   // wfa_hip_align_pair: the two sequences of the call, back to back
  // lifetime: batches keep a pointer to their aligner; wfa_hip_destroy with batches still alive only marks the handle,
  // the last batch to go frees it
  int live_batches = 0;
  bool destroy_pending = false;
  // the workspace below is shared by every run of this aligner: a run enqueued on another stream than the previous
  // one first waits for ws_event (recorded after each run), so runs are stream-ordered whatever streams callers pass
  hipEvent_t ws_event = nullptr;
  hipStream_t ws_last_stream = nullptr;
  // the walks of a split stage's launch run on this stream, under the alignment kernel of the next launch (which writes
  // the other half of the workspace); created on first use
  hipStream_t side_stream = nullptr;
  // round 6: what ONE launch of a split stage hands on is aligned by the stages behind it on this stream, beside the split stage's
  // next launch (batch_run_once: "pipelined tail"); created on first use
  hipStream_t tail_stream = nullptr;
  hipEvent_t tail_fork[2] = {nullptr, nullptr}, tail_join = nullptr;
  hipEvent_t band_event[4] = {nullptr, nullptr, nullptr, nullptr}, walk_event[4] = {nullptr, nullptr, nullptr, nullptr};   // [0..1] the band stages' walks, [2..3] the lane-full stage's expands
  // second upload stream of the host-packed upload (every other slot's DMAs: two copy engines)
  hipStream_t up_stream = nullptr;
  hipEvent_t up_fork = nullptr, up_join = nullptr;
  bool ws_event_recorded = false;
  // pinned staging ring of the pipelined upload (batches of >= 256 k pairs): host threads copy pieces of the caller's
  // pageable arrays into the slots, each slot goes to the device by DMA as soon as it is full
  std::vector<uint8_t*> pin_slot;
  std::vector<hipEvent_t> pin_ev;
  std::vector<char> pin_ev_recorded;   // the slot's event was recorded: its DMA must be over before the slot is refilled
  // single calls of a pywfa-style loop (a handful of pairs): one pinned, device-visible staging block + its device copy,
  // allocated once; the call is then host writes -> copy kernel -> alignment kernel -> one stream sync -> host reads
  uint8_t* tiny_h = nullptr;
  uint8_t* tiny_hd = nullptr;
  uint8_t* tiny_d = nullptr;
  // the resident one-pair kernel (round 6; wfa_slim.hpp: wfa_slim_kernel_mailbox): its mailbox in pinned host memory, the stream its
  // instances run on, the arguments the running instance was started with (another configuration / workspace: it is told to leave first)
  wfa::SlimMailbox* mb_h = nullptr;
  wfa::SlimMailbox* mb_d = nullptr;
  hipStream_t mb_stream = nullptr;
  wfa::BandArgs mb_args;
  bool mb_args_valid = false;
  uint32_t mb_seq = 0;
  int mb_failures = 0;
  size_t pin_slot_bytes = 0;
  int cu_count = 256;
  size_t total_mem = 0;
  std::string err;
  // persistent workspace for the general kernel (grown on demand)
  int32_t* ws = nullptr;
  size_t ws_bytes = 0;
  // device blocks of finished batches kept for the next one (a batch takes ~20 arrays; for the small batches of
  // a pywfa-style loop of single alignments hipMalloc / hipFree are most of the call)
  std::multimap<size_t, void*> pool_free;
  std::unordered_map<void*, size_t> pool_size;
  size_t pool_cached = 0;
};

static inline int knob(const wfa_hip_aligner* al, WfaKnob k, int dflt) {
  return al->knobs.set[k] ? al->knobs.value[k] : dflt;
}

// Blocks of finished batches are kept for the next batch: small ones (a pywfa-style loop of single alignments) and the
// large arrays of a big batch alike (a loop of wfa_hip_align_batch calls over equally shaped batches then allocates
// nothing: hipFree of a 10 M-pair batch's arrays alone was 23 ms of a 133 ms call).  At most 8 GB stay cached.
static const size_t POOL_MAX_BLOCK = (size_t)4 << 30, POOL_MAX_CACHED = (size_t)8 << 30;

// frees the cached (idle) blocks only; blocks in use by live batches keep their entries in pool_size
static void pool_drain_cached(wfa_hip_aligner* al) {
  for (auto& kv : al->pool_free) { al->pool_size.erase(kv.second); (void)hipFree(kv.second); }
  al->pool_free.clear(); al->pool_cached = 0;
}

static hipError_t pool_alloc(wfa_hip_aligner* al, void** p, size_t bytes) {
  size_t want = 256;
  if (bytes > POOL_MAX_BLOCK) want = (bytes + 255) & ~(size_t)255;
  else while (want < bytes) want <<= 1;
  if (want <= POOL_MAX_BLOCK) {
    auto it = al->pool_free.find(want);
    if (it != al->pool_free.end()) {
      *p = it->second; al->pool_cached -= want; al->pool_free.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, want);
  if (e == hipErrorOutOfMemory && al->pool_cached > 0) {
    // the cache holds idle blocks of other sizes: give them back and try once more
    (void)hipGetLastError();
    pool_drain_cached(al);
    e = hipMalloc(p, want);
  }
  if (e == hipSuccess) al->pool_size[*p] = want;
  return e;
}

static void pool_release(wfa_hip_aligner* al, void* p) {
  if (!p) return;
  auto it = al->pool_size.find(p);
  if (it == al->pool_size.end()) { (void)hipFree(p); return; }
  const size_t sz = it->second;
  if (sz <= POOL_MAX_BLOCK && al->pool_cached + sz <= POOL_MAX_CACHED) { al->pool_free.emplace(sz, p); al->pool_cached += sz; return; }
  al->pool_size.erase(it);
  (void)hipFree(p);
}

static void pool_drain(wfa_hip_aligner* al) {
  for (auto& kv : al->pool_free) (void)hipFree(kv.second);
  al->pool_free.clear(); al->pool_size.clear(); al->pool_cached = 0;
}

struct wfa_hip_batch {
  wfa_hip_aligner* al = nullptr;
  // configuration in force when the batch was created: the layout of the batch (op regions, 8-bit work list, checked
  // free ends) follows it, so run / sync / results use this snapshot, never the aligner's current configuration
  wfa_hip_config_t cfg;
  WfaDevConfig dcfg;
  WfaDevConfig gcfg;       // the general kernel's configuration (= dcfg unless dcfg.lin: the original one-component distance)
  int gncomp = 3;
  int wild = -1;           // the wildcard letter of the 8-bit pairs when dcfg.wildcard was cleared for the 2-bit ones (round 5, below)
  int ncomp = 3;
  int64_t n = 0;
  // host copies needed later
  std::vector<int32_t> h_plen, h_tlen;
  std::vector<int64_t> h_coff;
  std::unique_ptr<WfaPairMeta[]> h_meta;  // kept until the batch dies: its upload may still be in flight when batch_build returns
  std::vector<wfa::WfaPieceDesc> h_pieces;   // host-packed upload with 16-bit lengths: the pieces' first pair / first word (uploaded; kept like h_meta)
  uint32_t* d_len16 = nullptr;               // ... the {plen, tlen} halves as uploaded, and the piece table on the device
  wfa::WfaPieceDesc* d_pieces = nullptr;
  int max_width = 0;       // max(plen+tlen)+3
  int max_len = 0;         // max(plen, tlen)
  int64_t packed_bytes = 0;  // sum of ceil(len/4) over all sequences (algorithmic 2-bit bytes)
  int64_t ops_bytes = 0;     // sum(plen+tlen)
  // device
  uint8_t* d_bytes = nullptr;
  int64_t* d_pboff = nullptr;
  int64_t* d_tboff = nullptr;
  WfaPairMeta* d_meta = nullptr;
  uint32_t* d_words = nullptr;
  uint8_t* d_flags = nullptr;
  int32_t* d_score = nullptr;
  int32_t* d_status = nullptr;
  uint8_t* d_ops = nullptr;
  int64_t* d_cigar_off = nullptr;
  int64_t* d_cigar_begin = nullptr;
  int32_t* d_cigar_len = nullptr;
  uint32_t* d_list_packed = nullptr;  // worklists (nullptr = identity over all pairs)
  uint32_t* d_list_bytes = nullptr;
  uint32_t n_packed = 0, n_bytes = 0;
  uint32_t* d_fb_list2[2] = {nullptr, nullptr};  // leftover lists handed from one kernel stage to the next (ping-pong)
  const uint32_t* leftover_count = nullptr;       // device count of the pairs that reached the general kernel
  uint32_t* d_ovf_list[2] = {nullptr, nullptr};  // pairs whose arena overflowed
  uint32_t* d_counters = nullptr;  // [0] fallback count, [1] overflow count A, [2] overflow count B, [4..5] the pilots, [8..15] lane-full list / debug, [16..] one hand-over count per stage of a run
  std::vector<hipEvent_t> ev;   // 2 events per run since the last sync (kernel timing)
  size_t ev_used = 0;
  int runs_pending = 0;
  double ms_sum = 0.0; int ms_runs = 0;
  bool ran = false, synced = true;
  float last_ms = 0.f;
  int64_t last_kernel_pairs = 0;
  int64_t last_fallback = 0;
  hipStream_t last_stream = nullptr;
  bool uploads_pending = false;
  // recorded on the aligner's stream after the last upload / memset / pack kernel of batch_build: a run on another stream
  // waits for it (the host-packed upload returns with its DMAs still in flight)
  hipEvent_t upload_event = nullptr;
  int stage_pick = 0;  // first register-kernel stage chosen by the pilot of the first run (0 = not yet): 16, 32 or 64 lanes
  int segh_pick = 0;   // the same for the general form of the 32-lane segments (wfa_seg_kernel<.., HEUR>)
  int band_pick = 0;   // exact reads of 300 - 1 200 bases: the 256-diagonal register window first (1) or not (2: its pilot handed on most pairs), 0 undecided
  int laneh_pick = 0;  // general score-only form of the lane kernel first (wf-adaptive / free ends / step limit): 1 yes, 2 no (its pilot), 0 undecided
  int64_t arena_ints = 0;  // FULL: arena size used by the last launch (the part that grows 8x when a pair overflows it)
  int64_t arena_fixed = 0; // FULL, piggy-back history of the general kernel: the score-only ring in front of the growing part
  // device-side result surface (RLE)
  int32_t* d_plen = nullptr; int32_t* d_tlen = nullptr; int32_t* d_run_count = nullptr; int32_t* d_locs = nullptr;
  int64_t* d_run_off = nullptr; int64_t rle_total = -1;
};

#define HIP_TRY(al, expr)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      char buf_[512];                                                                          \
      snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      if (al) (al)->err = buf_;                                                                \
      g_error = buf_;                                                                          \
      return WFA_HIP_EDEVICE;                                                                  \
    }                                                                                          \
  } while (0)

// ------------------------------------------------------------------------------------------------
// configuration
// ------------------------------------------------------------------------------------------------
extern "C" int wfa_hip_abi_version(void) { return WFA_HIP_ABI_VERSION; }

extern "C" int wfa_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { g_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return WFA_HIP_EDEVICE; }
  return n;
}

extern "C" const char* wfa_hip_global_error(void) { return g_error.c_str(); }

extern "C" int wfa_hip_config_default(wfa_hip_config_t* c) {
  if (!c) return WFA_HIP_EINVAL;
  memset(c, 0, sizeof(*c));
  c->distance = WFA_DIST_AFFINE;  // align.pyx:311
  c->match = 0; c->mismatch = 4; c->gap_opening = 6; c->gap_extension = 2;  // align.pyx:313-316
  c->gap_opening2 = 24; c->gap_extension2 = 1;                               // align.pyx:317-318
  c->scope = WFA_SCOPE_FULL;      // align.pyx:319
  c->span = WFA_SPAN_ENDSFREE;    // align.pyx:320
  c->heuristic = WFA_HEUR_NONE;   // align.pyx:325
  c->min_wavefront_length = 10; c->max_distance_threshold = 50; c->steps_between_cutoffs = 1;  // :328-330
  c->xdrop = 20;                  // align.pyx:331
  c->memory_mode = WFA_MEM_HIGH;  // align.pyx:312
  c->max_steps = 0;               // align.pyx:333
  c->wildcard = -1;               // align.pyx:332
  return WFA_HIP_OK;
}

static int fail_cfg(char* err, size_t errlen, int code, const char* msg) {
  if (err && errlen) snprintf(err, errlen, "%s", msg);
  g_error = msg;
  return code;
}

extern "C" int wfa_hip_config_validate(const wfa_hip_config_t* c, char* err, size_t errlen) {
  if (!c) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "null config");
  if (c->distance < WFA_DIST_INDEL || c->distance > WFA_DIST_AFFINE2P)
    return fail_cfg(err, errlen, WFA_HIP_EINVAL, "unknown distance");
  // wavefront_penalties.c:65-173 (the reference prints and exit(1)s)
  if (c->distance == WFA_DIST_LINEAR) {
    if (c->match > 0) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "Match score must be negative or zero");
    if (c->mismatch <= 0 || c->gap_extension <= 0) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "Penalties must be (X>0,D>0,I>0)");
  } else if (c->distance >= WFA_DIST_AFFINE) {
    if (c->match > 0) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "Match score must be negative or zero");
    if (c->mismatch <= 0 || c->gap_opening < 0 || c->gap_extension <= 0)
      return fail_cfg(err, errlen, WFA_HIP_EINVAL, "Penalties must be (X>0,O>=0,E>0)");
  } else if (c->heuristic == WFA_HEUR_XDROP) {
    // wavefront_align.c:80-85
    return fail_cfg(err, errlen, WFA_HIP_EINVAL, "Heuristics drops are not compatible with 'edit'/'indel' distance metrics");
  }
  if (c->distance == WFA_DIST_AFFINE2P && (c->gap_opening2 < 0 || c->gap_extension2 <= 0))
    return fail_cfg(err, errlen, WFA_HIP_EINVAL, "Penalties must be (X>0,O1>=0,E1>0,O2>=0,E2>0)");
  if (c->scope != WFA_SCOPE_SCORE && c->scope != WFA_SCOPE_FULL) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "unknown scope");
  if (c->span != WFA_SPAN_END2END && c->span != WFA_SPAN_ENDSFREE) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "unknown span");
  if (c->heuristic < WFA_HEUR_NONE || c->heuristic > WFA_HEUR_XDROP) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "unknown heuristic");
  if (c->memory_mode < WFA_MEM_HIGH || c->memory_mode > WFA_MEM_BIWFA) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "unknown memory_mode");
  if (c->memory_mode == WFA_MEM_BIWFA) {
    // BiWFA (R/wavefront_bialign.c).  scope=score: wavefront_bialign_compute_score (:662-702) returns the score of the optimal
    // alignment, which is what the other memory modes return (checked on every metric against the real library,
    // the CPU checkers under tests/), and the score-only kernels hold O(s) state already; with a step limit (counted over the
    // forward + reverse scores, :475,513) the top-level breakpoint search itself runs on the device.  scope=full: the breakpoint
    // recursion on the device (csrc/wfa_biwfa.hpp), step limit included.  Round 4: with a heuristic too — the forward and the reverse
    // aligner of every breakpoint search cut their wavefronts off (R/wavefront_bialigner.c:53,161-166), also for scope=score (the
    // bidirectional cut-offs are not the unidirectional ones).  Not built: free ends (the reference itself exit(1)s,
    // R/wavefront_align.c:60-75).
    const bool free_ends = c->span == WFA_SPAN_ENDSFREE &&
                           (c->pattern_begin_free | c->pattern_end_free | c->text_begin_free | c->text_end_free) != 0;
    if (free_ends)
      return fail_cfg(err, errlen, WFA_HIP_ENOTSUP, "memory_mode biwfa is on the accelerated path without free ends only");
  }
  if (c->pattern_begin_free < 0 || c->pattern_end_free < 0 || c->text_begin_free < 0 || c->text_end_free < 0)
    return fail_cfg(err, errlen, WFA_HIP_EINVAL, "ends-free sizes must be >= 0");
  // match < 0 with free begins (the ends-free re-seeding of R/wavefront_compute.c:124-254): score scope.  With a backtrace the
  // reference itself fails on ordinary inputs (exit(-1) "I?/D?-Beginning backtrace error" in memory mode high, an endless loop
  // in medium / low; reproducers among the CPU tests), so scope=full has nothing to be equal to
  if (c->distance >= WFA_DIST_LINEAR && c->match < 0 && c->span == WFA_SPAN_ENDSFREE && (c->pattern_begin_free > 0 || c->text_begin_free > 0) &&
      c->scope == WFA_SCOPE_FULL)
    return fail_cfg(err, errlen, WFA_HIP_ENOTSUP, "match<0 with free begins is built for scope=score (with a backtrace the reference itself exits or hangs)");
  if (c->wildcard < -1 || c->wildcard > 255) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "wildcard must be -1 or a byte");
  if (c->reserved != 0) return fail_cfg(err, errlen, WFA_HIP_EINVAL, "reserved must be 0");
  return WFA_HIP_OK;
}

// Configurations that are ordinary gap-affine alignments in disguise run on the fast kernels, with the score translated afterwards
// (VERDICT r03 item 7):
//  * match < 0: the reference aligns with the Eizenga-rescaled penalties and reports (-match (plen + tlen) - s) / 2
//    (R/wavefront_penalties.c:95-173, R/wavefront_penalties.h:73, R/wavefront_compute.c:108-120): here match = 0 + the rescaled
//    penalties + score_mode 1;
//  * indel / levenshtein / gap-linear in SCORE scope: a gap-linear recurrence (R/wavefront_compute_linear.c:44-74,
//    R/wavefront_compute_edit.c:44-100) gives the same scores as gap-affine with o = 0 (I[s][k] = max(M, I)[s-e][k-1] + 1 = M[s-e][k-1] + 1
//    because M >= I), and the indel distance the same as mismatch = 2 (a mismatch is an insertion + a deletion): gap-affine
//    (x, 0, indel) + score_mode 2 for the two distances reported as +s.  Op strings can differ where the backtraces break ties
//    differently (R/wavefront_backtrace.c:223-319 vs :320-529), so scope = full keeps the one-component kernel.
// Only where nothing else reads the original form: no free ends (the reference re-seeds free begins under match < 0,
// R/wavefront_compute.c:124-254; its partial-alignment score uses the end position), no X-drop (its score uses match), no step
// limit (-max_steps is reported in the original scale), no wildcard, not BiWFA.  WFA_HIP_NO_SCORE_MAP=1: off.
static void map_to_gap_affine(const wfa_hip_config_t& c, WfaDevConfig* d, int* ncomp) {
  d->score_mode = 0; d->sw_match = 0;
  static const bool off = getenv("WFA_HIP_NO_SCORE_MAP") && *getenv("WFA_HIP_NO_SCORE_MAP") == '1';
  if (off) return;
  const bool free_ends = c.span == WFA_SPAN_ENDSFREE && (c.pattern_begin_free | c.pattern_end_free | c.text_begin_free | c.text_end_free) != 0;
  if (free_ends || c.max_steps > 0 || c.wildcard >= 0 || c.memory_mode == WFA_MEM_BIWFA) return;
  if (c.heuristic != WFA_HEUR_NONE && !(c.heuristic == WFA_HEUR_ADAPTIVE && *ncomp != 1)) return;
  if (*ncomp == 1) {
    // Round 6, with CIGARs: gap-linear and levenshtein (match = 0) take the register kernels' LIN form (wfa_lane.hpp: gap-affine with o = 0 and
    // no extension candidates = the one-component recurrences and the linear backtrace's choices); what those stages hand on goes to the
    // general kernel under the ORIGINAL configuration (derive_dev_config keeps it).  indel: the same without the mismatch candidate.
    if (c.scope != WFA_SCOPE_SCORE) {
      static const bool lin_off = getenv("WFA_HIP_NO_LIN") && *getenv("WFA_HIP_NO_LIN") == '1';
      if (lin_off || d->match != 0 || !d->rtc) return;
      d->lin = (c.distance == WFA_DIST_INDEL) ? 2 : 1;   // (2: no mismatch candidate)
    }
    const int indel = d->o1;   // (derive_dev_config keeps the indel penalty there)
    if (c.distance == WFA_DIST_INDEL) d->x = 2;
    d->o1 = 0; d->e1 = indel; d->o2 = 0; d->e2 = indel;
    d->metric = WFA_DIST_AFFINE;
    *ncomp = 3;
    d->scope = std::max(d->o1 + d->e1, d->x) + 1;
    d->score_mode = (c.distance == WFA_DIST_LINEAR) ? (d->match < 0 ? 1 : 0) : 2;
  } else {
    if (d->match == 0) return;
    d->score_mode = 1;
  }
  if (d->match < 0) { d->sw_match = -d->match; d->match = 0; }
}

// gd / gncomp: the configuration the GENERAL kernel runs — the same, unless a one-component distance with CIGARs was mapped (d->lin)
static void derive_dev_config(const wfa_hip_config_t& c, WfaDevConfig* d, int* ncomp, WfaDevConfig* gd = nullptr, int* gncomp = nullptr) {
  const bool two = (c.distance == WFA_DIST_AFFINE2P);
  d->lin = 0;
  d->metric = c.distance;
  if (c.distance <= WFA_DIST_LINEAR) {
    // single-component metrics: wavefront_penalties.c:39-94, wavefront_components.c:43-74
    *ncomp = 1;
    d->match = 0; d->x = 1; d->o1 = 1;
    if (c.distance == WFA_DIST_LINEAR) {
      if (c.match < 0) { d->match = c.match; d->x = 2 * c.mismatch - 2 * c.match; d->o1 = 2 * c.gap_extension - c.match; }
      else { d->x = c.mismatch; d->o1 = c.gap_extension; }  // pywfa passes gap_extension as the indel penalty (align.pyx:351-355)
    }
    d->e1 = 0; d->o2 = d->o1; d->e2 = 0;
    d->scope = (c.distance == WFA_DIST_LINEAR) ? std::max(d->x, d->o1) + 1 : 2;
    d->endsfree = (c.span == WFA_SPAN_ENDSFREE);
    d->pbf = c.pattern_begin_free; d->pef = c.pattern_end_free; d->tbf = c.text_begin_free; d->tef = c.text_end_free;
    d->heuristic = c.heuristic;
    d->min_wf_len = c.min_wavefront_length; d->max_dist_thr = c.max_distance_threshold;
    d->steps_between = c.steps_between_cutoffs; d->xdrop = c.xdrop;
    d->max_steps = (c.max_steps <= 0) ? INT_MAX : c.max_steps;
    d->wildcard = c.wildcard;
    d->biwfa_top = 0;
    d->rtc = wfa::rtc_enabled() ? 1 : 0;
    const WfaDevConfig orig1 = *d;
    map_to_gap_affine(c, d, ncomp);
    if (gd) { *gd = d->lin ? orig1 : *d; *gncomp = d->lin ? 1 : *ncomp; }
    return;
  }
  *ncomp = two ? 5 : 3;
  // Eizenga rescaling when match<0 (wavefront_penalties.c:113-122,148-164)
  if (c.match < 0) {
    d->match = c.match;
    d->x = 2 * c.mismatch - 2 * c.match;
    d->o1 = 2 * c.gap_opening; d->e1 = 2 * c.gap_extension - c.match;
    d->o2 = 2 * c.gap_opening2; d->e2 = 2 * c.gap_extension2 - c.match;
  } else {
    d->match = 0;
    d->x = c.mismatch;
    d->o1 = c.gap_opening; d->e1 = c.gap_extension;
    d->o2 = c.gap_opening2; d->e2 = c.gap_extension2;
  }
  if (!two) { d->o2 = d->o1; d->e2 = d->e1; }
  int scope_indel = d->o1 + d->e1;
  if (two) scope_indel = std::max(scope_indel, d->o2 + d->e2);
  d->scope = std::max(scope_indel, d->x) + 1;  // wavefront_components.c:81-124
  d->endsfree = (c.span == WFA_SPAN_ENDSFREE);
  d->pbf = c.pattern_begin_free; d->pef = c.pattern_end_free;
  d->tbf = c.text_begin_free; d->tef = c.text_end_free;
  d->heuristic = c.heuristic;
  d->min_wf_len = c.min_wavefront_length; d->max_dist_thr = c.max_distance_threshold;
  d->steps_between = c.steps_between_cutoffs; d->xdrop = c.xdrop;
  d->max_steps = (c.max_steps <= 0) ? INT_MAX : c.max_steps;  // align.pyx:415-417
  d->wildcard = c.wildcard;
  d->biwfa_top = 0;
  // penalty shapes without an instantiation: the register kernels are compiled for them at run time where hipRTC works (probed
  // when such a shape first asks: seg_shape / band_supported)
  d->rtc = wfa::rtc_enabled() ? 1 : 0;
  const WfaDevConfig orig = *d;
  const int orig_ncomp = *ncomp;
  map_to_gap_affine(c, d, ncomp);
  if (gd) { *gd = d->lin ? orig : *d; *gncomp = d->lin ? orig_ncomp : *ncomp; }
}

extern "C" wfa_hip_aligner_t* wfa_hip_create(const wfa_hip_config_t* cfg, int device) {
  char msg[256];
  if (wfa_hip_config_validate(cfg, msg, sizeof(msg)) != WFA_HIP_OK) return nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return nullptr;
  }
  if (device < 0 || device >= ndev) { g_error = "device ordinal out of range"; return nullptr; }
  if ((e = hipSetDevice(device)) != hipSuccess) { g_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
  wfa_hip_aligner* al = new wfa_hip_aligner();
  al->device = device;
  al->cfg = *cfg;
  derive_dev_config(al->cfg, &al->dcfg, &al->ncomp, &al->gcfg, &al->gncomp);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { g_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); delete al; return nullptr; }
  al->cu_count = prop.multiProcessorCount;
  al->total_mem = prop.totalGlobalMem;
  if ((e = hipStreamCreateWithFlags(&al->stream, hipStreamNonBlocking)) != hipSuccess) { g_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete al; return nullptr; }
  if ((e = hipEventCreateWithFlags(&al->ws_event, hipEventDisableTiming)) != hipSuccess) { g_error = std::string("hipEventCreate: ") + hipGetErrorString(e); (void)hipStreamDestroy(al->stream); delete al; return nullptr; }
  // Round 6: the aligner's other streams (walks, the pipelined tail, the second upload engine) are created HERE, back to back, and each is
  // given a first command at once.  The runtime spreads streams over a handful of hardware queues as they first submit work; created
  // lazily, in the middle of a process that has opened and closed other aligners, the main and the side stream of an aligner could land
  // on the SAME hardware queue — the walks of a launch then ran after the next alignment kernel instead of under it (bench.py's C3 after
  // the C2 / C1 legs: 44.7 ms where a fresh process measured 40.0; rocprofv3 showed both on queue 4).  Four streams bound in a row take
  // four different queues whatever was bound before.
  {
    // (the second upload stream stays lazy — WFA_HIP_EAGER_UP=1 binds it here too: bound in this row the ASCII upload of 10 M pairs
    // measured 93 ms instead of 30 ms)
    static const bool eager_up = getenv("WFA_HIP_EAGER_UP") && *getenv("WFA_HIP_EAGER_UP") == '1';
    hipStream_t* extra[3] = {&al->side_stream, &al->tail_stream, eager_up ? &al->up_stream : nullptr};
    bool ok = true;
    for (hipStream_t* sp : extra) ok = ok && (sp == nullptr || hipStreamCreateWithFlags(sp, hipStreamNonBlocking) == hipSuccess);
    hipStream_t all[4] = {al->stream, al->side_stream, al->tail_stream, al->up_stream};
    for (hipStream_t st : all) ok = ok && (st == nullptr || hipEventRecord(al->ws_event, st) == hipSuccess);
    if (!ok) {
      g_error = std::string("hipStreamCreate: ") + hipGetErrorString(hipGetLastError());
      for (hipStream_t st : all) if (st) (void)hipStreamDestroy(st);
      (void)hipEventDestroy(al->ws_event);
      delete al; return nullptr;
    }
  }
  al->knobs.load();
  {
    // one process per GPU (bench.py --gpus N under torch.distributed.run, any launcher that exports LOCAL_WORLD_SIZE): the ranks of the
    // node share its cores; WFA_HIP_HOST_SHARE says so explicitly
    const char* hs = getenv("WFA_HIP_HOST_SHARE");
    if (!(hs && *hs)) hs = getenv("LOCAL_WORLD_SIZE");
    al->host_share = (hs && atoi(hs) > 0) ? std::min(atoi(hs), 64) : 1;
  }
  // the process's CPUs as they are now, before any upload worker is bound (numa_lookup must not read a mask this library narrowed)
  al->proc_cpus_valid = sched_getaffinity(getpid(), sizeof(al->proc_cpus), &al->proc_cpus) == 0;
  return al;
}

static void mailbox_quit(wfa_hip_aligner* al);
static void aligner_free(wfa_hip_aligner* al) {
  (void)hipSetDevice(al->device);
  if (al->ws) (void)hipFree(al->ws);
  pool_drain(al);
  if (al->ws_event) (void)hipEventDestroy(al->ws_event);
  for (uint8_t* p : al->pin_slot) (void)hipHostFree(p);
  if (al->mb_h) { mailbox_quit(al); (void)hipHostFree(al->mb_h); }
  if (al->mb_stream) (void)hipStreamDestroy(al->mb_stream);
  if (al->tiny_h) (void)hipHostFree(al->tiny_h);
  if (al->tiny_d) (void)hipFree(al->tiny_d);
  for (hipEvent_t e : al->pin_ev) (void)hipEventDestroy(e);
  if (al->up_stream) (void)hipStreamDestroy(al->up_stream);
  if (al->up_fork) (void)hipEventDestroy(al->up_fork);
  if (al->up_join) (void)hipEventDestroy(al->up_join);
  if (al->side_stream) (void)hipStreamDestroy(al->side_stream);
  if (al->tail_stream) (void)hipStreamDestroy(al->tail_stream);
  for (int i = 0; i < 2; ++i) if (al->tail_fork[i]) (void)hipEventDestroy(al->tail_fork[i]);
  if (al->tail_join) (void)hipEventDestroy(al->tail_join);
  for (int i = 0; i < 4; ++i) { if (al->band_event[i]) (void)hipEventDestroy(al->band_event[i]); if (al->walk_event[i]) (void)hipEventDestroy(al->walk_event[i]); }
  if (al->stream) (void)hipStreamDestroy(al->stream);
  delete al;
}

extern "C" void wfa_hip_destroy(wfa_hip_aligner_t* al) {
  if (!al) return;
  // resident batches keep a pointer to their aligner (pool, stream, workspace): while any is alive the handle is only
  // marked, and the last wfa_hip_batch_destroy frees it
  if (al->live_batches > 0) { al->destroy_pending = true; return; }
  aligner_free(al);
}

extern "C" int wfa_hip_set_config(wfa_hip_aligner_t* al, const wfa_hip_config_t* cfg) {
  if (!al) return WFA_HIP_EINVAL;
  char msg[256];
  const int rc = wfa_hip_config_validate(cfg, msg, sizeof(msg));
  if (rc != WFA_HIP_OK) { al->err = msg; return rc; }
  al->cfg = *cfg;
  derive_dev_config(al->cfg, &al->dcfg, &al->ncomp, &al->gcfg, &al->gncomp);
  return WFA_HIP_OK;
}

extern "C" int wfa_hip_get_config(const wfa_hip_aligner_t* al, wfa_hip_config_t* cfg) {
  if (!al || !cfg) return WFA_HIP_EINVAL;
  *cfg = al->cfg;
  return WFA_HIP_OK;
}

extern "C" const char* wfa_hip_last_error(const wfa_hip_aligner_t* al) { return al ? al->err.c_str() : g_error.c_str(); }

// ------------------------------------------------------------------------------------------------
// batches
// ------------------------------------------------------------------------------------------------
static void batch_free(wfa_hip_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->al->device);
  void* ptrs[] = {b->d_bytes, b->d_pboff, b->d_tboff, b->d_meta, b->d_words, b->d_flags, b->d_score, b->d_status,
                  b->d_ops, b->d_cigar_off, b->d_cigar_begin, b->d_cigar_len, b->d_list_packed, b->d_list_bytes,
                  b->d_fb_list2[0], b->d_fb_list2[1], b->d_ovf_list[0], b->d_ovf_list[1], b->d_counters,
                  b->d_plen, b->d_tlen, b->d_run_count, b->d_locs, b->d_run_off, b->d_len16, b->d_pieces};
  // blocks go back to the aligner's pool: nothing of this batch may still be running
  const bool timing = b->al->knobs.set[K_TIMING];
  const double t0 = timing ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
  if (b->ran && !b->synced && b->last_stream) (void)hipStreamSynchronize(b->last_stream);
  (void)hipStreamSynchronize(b->al->stream);
  const double t1 = timing ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
  for (void* p : ptrs) pool_release(b->al, p);
  for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
  if (b->upload_event) (void)hipEventDestroy(b->upload_event);
  const double t2 = timing ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
  wfa_hip_aligner* al = b->al;
  delete b;
  if (timing) fprintf(stderr, "[wfa_hip] batch_free: sync %.3f ms, release %.3f ms, host free %.3f ms\n", t1 - t0, t2 - t1,
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t2);
  if (--al->live_batches == 0 && al->destroy_pending) aligner_free(al);
}

extern "C" void wfa_hip_batch_destroy(wfa_hip_batch_t* b) { batch_free(b); }

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Pipelined host -> device upload of pageable arrays: the arrays are cut into pieces of one staging
// slot; `nthreads` host threads claim pieces in order, copy them into the pinned ring and enqueue the DMA of their slot
// on `stream` themselves; a slot is refilled once its previous DMA is over.  Host copy and DMA overlap, and the DMA runs
// from pinned memory at the link's rate instead of the runtime's pageable path.  Returns after every piece is ENQUEUED.
// (src == nullptr: the piece is produced by gen(out, byte offset within the job, bytes) straight into the pinned slot)
struct UploadJob { void* dst; const void* src; size_t bytes; std::function<void(uint8_t*, size_t, size_t)> gen; };

static size_t staged_slot_bytes(const wfa_hip_aligner* al) { return (size_t)std::max(1, knob(al, K_PIPE_CHUNK, 8)) << 20; }

// Host threads of one device's upload pipeline.  The host is shared by every aligner / process that feeds a GPU of the node
// (al->host_share: the devices of a wfa_hip_multi_t, the ranks of a one-process-per-GPU job), so each takes its share of the logical
// CPUs: 8 devices on 256 CPUs pack with 16 threads and copy with 8 each (24 x 8 = 192 threads), one device alone with 32 + 8 as
// before (8 x (32 + 8) = 320 threads on 256 CPUs).  WFA_HIP_PACK_THREADS / WFA_HIP_PIPE_THREADS override.
extern "C" int wfa_hip_plan_host_threads(int sharers, int hw_threads, int* pack_threads, int* copy_threads) {
  if (sharers < 1 || hw_threads < 1 || !pack_threads || !copy_threads) return WFA_HIP_EINVAL;
  const int budget = std::max(1, hw_threads / sharers);          // logical CPUs of one sharer
  *pack_threads = std::max(1, std::min(32, budget / 2));          // (packing is compute: more threads than the plain copy needs)
  *copy_threads = std::max(1, std::min(8, budget / 4));
  return WFA_HIP_OK;
}
static int staged_copy_threads(const wfa_hip_aligner* al) {
  int pack = 1, copy = 1;
  (void)wfa_hip_plan_host_threads(std::max(1, al->host_share), std::max(1, (int)std::thread::hardware_concurrency()), &pack, &copy);
  return al->knobs.set[K_PIPE_THREADS] ? std::max(1, std::min(knob(al, K_PIPE_THREADS, 8), (int)std::thread::hardware_concurrency())) : copy;
}
static int staged_pack_threads(const wfa_hip_aligner* al) {
  int pack = 1, copy = 1;
  (void)wfa_hip_plan_host_threads(std::max(1, al->host_share), std::max(1, (int)std::thread::hardware_concurrency()), &pack, &copy);
  return al->knobs.set[K_PACK_THREADS] ? std::max(1, std::min(knob(al, K_PACK_THREADS, 32), std::max(1, (int)std::thread::hardware_concurrency() / 2))) : pack;
}

// the ring serves both forms of the upload: sized for the larger team so that alternating calls do not re-allocate it
// The pinned ring is allocated with hipHostMallocDefault: without hipHostMallocNumaUser HIP places pinned host memory on the NUMA node
// closest to the current device.  Round 5 bound the threads that fill it (2-bit packing, copies) to that node's CPUs whatever the
// call; the driver's run of round 5 then measured 187 M aln/s where round 4 had 332 M: a packer READS 300 B of the caller's pages per
// pair and WRITES 92 B into the ring, so where the caller's pages live matters three times more than where the ring lives, and on
// a box whose GPU hangs off the other socket every bound packer pulled its input across the socket link.  Measured in round 6
// (tools/probes/e2e_numa.py, profiles/r06_e2e_numa.txt: 2 x EPYC 9575F, GPU on node 0, 10 M x 150 bp, median of 7 calls):
//   input on the GPU's node:   never bound 323 M aln/s, always bound 301 M, bound because the input is local 303 M
//   input on the other node:   never bound 271 M,       always bound 228 M, auto (= not bound) 278 M
// Binding never won: the upload is bound by the DMA (0.96 GB at ~46 GB/s; 16 to 64 packing threads measure the same), and threads
// spread over both sockets reach it with room to spare.  So the default is NOT to bind (WFA_HIP_NUMA unset or "0" = round 4's
// behaviour); "auto" binds the spawned workers to the device's node only when the kernel says the caller's sequences live there
// (get_mempolicy on five sampled pages), "1" always (round 5's behaviour) — both kept for hosts where the socket link is the
// scarcer resource (eight GPUs uploading at once).
// Only the threads this library spawns are ever bound; the caller's thread, which works a share of the pieces too, keeps its mask
// (binding it narrowed the application's main thread for good, and every thread it created afterwards).
// The node comes from sysfs (the device's PCI address), the CPUs are intersected with the process's affinity mask as it was when the
// aligner was created; anything missing = no binding.
static void numa_lookup(wfa_hip_aligner* al) {
  al->numa_state = 2;
  const char* off = getenv("WFA_HIP_NO_NUMA");
  if (off && *off == '1') return;
  const char* mode = getenv("WFA_HIP_NUMA");
  al->numa_mode = (mode && *mode == '1') ? 1 : (mode && *mode == 'a') ? 2 : 0;   // 1 always, 2 auto, 0 (default) never
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, al->device) != hipSuccess) { (void)hipGetLastError(); return; }
  for (char* c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
  auto read_line = [](const std::string& path, char* buf, size_t cap) -> bool {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)cap, f) != nullptr;
    fclose(f);
    return ok;
  };
  char line[4096];
  if (!read_line(std::string("/sys/bus/pci/devices/") + bus + "/numa_node", line, sizeof(line))) return;
  const int node = atoi(line);
  if (node < 0) return;
  al->numa_node = node;
  if (!read_line("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", line, sizeof(line))) return;
  cpu_set_t want;
  CPU_ZERO(&want);
  if (!al->proc_cpus_valid) return;
  for (const char* p = line; *p && *p != '\n';) {   // "0-63,128-191"
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0 && CPU_ISSET((int)c, &al->proc_cpus)) CPU_SET((int)c, &want);
    if (*p == ',') ++p;
  }
  if (CPU_COUNT(&want) < 8) return;   // (a container pinned to a few CPUs: leave the scheduler alone)
  if (CPU_EQUAL(&want, &al->proc_cpus)) return;   // (one node, or the process is confined to this node already: nothing to bind)
  al->numa_cpus = want; al->numa_state = 1;
}
// NUMA node of the page that holds `addr` (-1: unknown — no NUMA, the call is not permitted here, the page is not resident)
static int page_node(const void* addr) {
#if defined(SYS_get_mempolicy)
  int node = -1;
  if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, const_cast<void*>(addr), 3ul /* MPOL_F_NODE | MPOL_F_ADDR */) != 0) return -1;
  return node;
#else
  (void)addr; return -1;
#endif
}
// decide, for one call, whether the spawned upload workers are bound to the device's node: the input must live there (auto)
static bool upload_binds(wfa_hip_aligner* al, const uint8_t* src, size_t bytes) {
  al->last_src_node = -1; al->last_bound = 0;
  if (al->numa_state != 1 || al->numa_mode == 0) return false;
  if (al->numa_mode == 1) { al->last_bound = 1; return true; }
  if (!src || bytes == 0) return false;
  int votes = 0, seen = 0;
  for (int i = 0; i < 5; ++i) {   // five pages spread over the blob
    const int nd = page_node(src + (size_t)((double)bytes * (0.1 + 0.2 * i)));
    if (nd < 0) continue;
    ++seen; al->last_src_node = nd;
    if (nd == al->numa_node) ++votes;
  }
  const bool bind = seen > 0 && votes == seen;
  al->last_bound = bind ? 1 : 0;
  return bind;
}
static inline void bind_upload_worker(const wfa_hip_aligner* al, bool bind) {
  if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &al->numa_cpus);
}
extern "C" int wfa_hip_upload_info(const wfa_hip_aligner_t* al, int32_t* info, int n) {
  if (!al || !info || n < 8) return WFA_HIP_EINVAL;
  info[0] = al->numa_node; info[1] = al->numa_state == 1 ? CPU_COUNT(&al->numa_cpus) : 0; info[2] = al->numa_mode;
  info[3] = al->last_src_node; info[4] = al->last_bound; info[5] = staged_pack_threads(al); info[6] = staged_copy_threads(al);
  info[7] = al->proc_cpus_valid ? CPU_COUNT(&al->proc_cpus) : 0;
  return WFA_HIP_OK;
}

static int staged_ring(wfa_hip_aligner* al) {
  if (al->numa_state == 0) numa_lookup(al);
  const size_t slot_bytes = staged_slot_bytes(al);
  const int nslots = std::max(staged_copy_threads(al), staged_pack_threads(al)) + 4;
  if (al->pin_slot_bytes != slot_bytes || (int)al->pin_slot.size() != nslots) {
    for (uint8_t* p : al->pin_slot) (void)hipHostFree(p);
    for (hipEvent_t e : al->pin_ev) (void)hipEventDestroy(e);
    al->pin_slot.clear(); al->pin_ev.clear(); al->pin_ev_recorded.clear(); al->pin_slot_bytes = 0;
    for (int i = 0; i < nslots; ++i) {
      uint8_t* p = nullptr; hipEvent_t e;
      HIP_TRY(al, hipHostMalloc((void**)&p, slot_bytes, hipHostMallocDefault));
      al->pin_slot.push_back(p);
      HIP_TRY(al, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      al->pin_ev.push_back(e);
      al->pin_ev_recorded.push_back(0);
    }
    al->pin_slot_bytes = slot_bytes;
  }
  return WFA_HIP_OK;
}

static int staged_upload(wfa_hip_aligner* al, const std::vector<UploadJob>& jobs, hipStream_t stream) {
  const size_t slot_bytes = staged_slot_bytes(al);
  const int nthreads = staged_copy_threads(al);
  { const int rrc = staged_ring(al); if (rrc != WFA_HIP_OK) return rrc; }
  const int nslots = (int)al->pin_slot.size();
  struct Piece { void* dst; const uint8_t* src; size_t bytes; const UploadJob* job; size_t off; };
  std::vector<Piece> pieces;
  for (const UploadJob& j : jobs)
    for (size_t o = 0; o < j.bytes; o += slot_bytes)
      pieces.push_back({(uint8_t*)j.dst + o, j.src ? (const uint8_t*)j.src + o : nullptr, std::min(slot_bytes, j.bytes - o), &j, o});
  const long np = (long)pieces.size();
  std::atomic<long> next(0);
  std::vector<std::atomic<long>> issued((size_t)nslots);   // index of the last piece whose DMA was enqueued from this slot
  for (auto& x : issued) x.store(-1);
  std::atomic<int> failed(0);
  const int device = al->device;
  const bool bind = upload_binds(al, jobs.empty() ? nullptr : (const uint8_t*)jobs[0].src, jobs.empty() ? 0 : jobs[0].bytes);
  auto worker = [&](bool spawned) {
    bind_upload_worker(al, bind && spawned);   // (never the caller's own thread: ADVICE r05)
    (void)hipSetDevice(device);
    for (;;) {
      const long i = next.fetch_add(1);
      if (i >= np || failed.load()) return;
      const int sl = (int)(i % nslots);
      // the slot's previous piece (of this call, or of an earlier one): wait until its DMA was enqueued (by whichever
      // thread had it), then until it is over
      // (a worker that saw `failed` leaves without publishing its piece: whoever waits for that piece must leave too)
      if (i >= nslots) while (issued[(size_t)sl].load(std::memory_order_acquire) != i - nslots) { if (failed.load()) return; std::this_thread::yield(); }
      if (al->pin_ev_recorded[(size_t)sl] && hipEventSynchronize(al->pin_ev[(size_t)sl]) != hipSuccess) failed.store(1);
      if (pieces[(size_t)i].src) memcpy(al->pin_slot[(size_t)sl], pieces[(size_t)i].src, pieces[(size_t)i].bytes);
      else pieces[(size_t)i].job->gen(al->pin_slot[(size_t)sl], pieces[(size_t)i].off, pieces[(size_t)i].bytes);
      if (hipMemcpyAsync(pieces[(size_t)i].dst, al->pin_slot[(size_t)sl], pieces[(size_t)i].bytes, hipMemcpyHostToDevice, stream) != hipSuccess ||
          hipEventRecord(al->pin_ev[(size_t)sl], stream) != hipSuccess) failed.store(1);
      al->pin_ev_recorded[(size_t)sl] = 1;
      issued[(size_t)sl].store(i, std::memory_order_release);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, true);
  worker(false);
  for (auto& x : th) x.join();
  if (failed.load()) { al->err = "pipelined upload failed"; return WFA_HIP_EDEVICE; }
  return WFA_HIP_OK;
}

// The mirror image for large results (round 6: the op bytes of long reads — C5's 8 192 x 100 kb pairs are 1.6 GB): device -> pinned slot
// by DMA, pinned slot -> the caller's pageable array by the host threads, a few pieces in flight.  A plain hipMemcpy into pageable
// memory moves such a block at a few GB/s, and into a FRESH array (np.zeros: untouched pages) every page fault is taken by the one
// thread the runtime copies with; here the faults are spread over the team.  Returns when the caller's array is complete.
static int staged_download(wfa_hip_aligner* al, uint8_t* dst, const uint8_t* src_dev, size_t bytes, hipStream_t stream) {
  const size_t slot_bytes = staged_slot_bytes(al);
  { const int rrc = staged_ring(al); if (rrc != WFA_HIP_OK) return rrc; }
  const int nslots = (int)al->pin_slot.size();
  const long np = (long)((bytes + slot_bytes - 1) / slot_bytes);
  const int nthreads = (int)std::max<long>(1, std::min<long>(std::min<long>(staged_pack_threads(al), nslots - 2), np));
  std::atomic<long> next(0);
  std::vector<std::atomic<long>> drained((size_t)nslots);   // index of the last piece copied OUT of the slot
  for (auto& x : drained) x.store(-1);
  std::atomic<int> failed(0);
  std::mutex enq;   // DMAs are enqueued in piece order (a slot's event must be the one of its own piece)
  const int device = al->device;
  auto worker = [&]() {
    (void)hipSetDevice(device);
    for (;;) {
      const long i = next.fetch_add(1);
      if (i >= np || failed.load()) return;
      const int sl = (int)(i % nslots);
      const size_t off = (size_t)i * slot_bytes, nb = std::min(slot_bytes, bytes - off);
      if (i >= nslots) while (drained[(size_t)sl].load(std::memory_order_acquire) != i - nslots) { if (failed.load()) return; std::this_thread::yield(); }
      if (al->pin_ev_recorded[(size_t)sl] && hipEventSynchronize(al->pin_ev[(size_t)sl]) != hipSuccess) failed.store(1);   // (an upload that used the slot)
      {
        std::lock_guard<std::mutex> lock(enq);
        if (hipMemcpyAsync(al->pin_slot[(size_t)sl], src_dev + off, nb, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipEventRecord(al->pin_ev[(size_t)sl], stream) != hipSuccess) failed.store(1);
        al->pin_ev_recorded[(size_t)sl] = 1;
      }
      if (!failed.load() && hipEventSynchronize(al->pin_ev[(size_t)sl]) != hipSuccess) failed.store(1);
      if (!failed.load()) memcpy(dst + off, al->pin_slot[(size_t)sl], nb);
      drained[(size_t)sl].store(i, std::memory_order_release);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
  worker();
  for (auto& x : th) x.join();
  if (failed.load()) { (void)hipGetLastError(); al->err = "pipelined download failed"; return WFA_HIP_EDEVICE; }
  return WFA_HIP_OK;
}

// Host-packed form of the pipelined upload: the host threads pack the caller's ASCII sequences to
// 2 bits per base (host_pack.cpp: AVX-512 / AVX2 / plain C) straight into the pinned slots, together with the per-pair
// metadata, so 76 + 16 B per 150 bp pair cross PCIe instead of 300 + 32 B and the device pack kernel is not run.
// A piece is a run of whole 64-pair blocks whose words and metadata fit one slot.  Pairs with a letter outside ACGT are
// listed in `flagged` (ascending); the caller uploads their bytes separately.  Returns after every piece is ENQUEUED:
// the caller's arrays are not read after that.
namespace wfa { bool host_pack_seq(const uint8_t* s, int len, uint32_t* out, int form); void host_repack2_seq(const uint8_t* s, int len, uint32_t* out); }

struct PackPiece { int64_t lo, hi; uint64_t wlo; uint64_t nwords; };

static int staged_pack_upload(wfa_hip_aligner* al, wfa_hip_batch* b, const std::vector<PackPiece>& pieces, const uint8_t* seqs,
                              const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len,
                              hipStream_t stream, std::vector<uint32_t>* flagged, bool in2bit = false, uint32_t* d_len16 = nullptr) {
  // in2bit: the caller's sequences are 2-bit already (four bases per byte at BYTE offsets): re-based to whole words and re-coded here
  // d_len16: the metadata crosses PCIe as 4 B per pair {plen, tlen} (16-bit each) into d_len16; wfa_meta_from_len16_kernel rebuilds
  //          the 16 B records on the device (round 6: 96 -> 84 B per 150 bp pair)
  { const int rrc = staged_ring(al); if (rrc != WFA_HIP_OK) return rrc; }
  const int nslots = (int)al->pin_slot.size();
  const int nthreads = std::min<int>(staged_pack_threads(al), std::max<int>(1, (int)pieces.size()));
  const long np = (long)pieces.size();
  std::atomic<long> next(0);
  std::vector<std::atomic<long>> issued((size_t)nslots);
  for (auto& x : issued) x.store(-1);
  std::atomic<int> failed(0);
  std::vector<std::vector<uint32_t>> bad((size_t)np);
  const int device = al->device;
  // two upload streams: the DMAs of every other piece go to a second stream, forked from / joined into `stream` (two copy
  // engines: pack + upload of 10 M x 150 bp 23 -> 21 ms)
  const bool two_up = knob(al, K_UP_STREAMS, 2) >= 2 && np >= 4;
  if (two_up) {
    if (!al->up_stream) HIP_TRY(al, hipStreamCreateWithFlags(&al->up_stream, hipStreamNonBlocking));
    if (!al->up_fork) {
      HIP_TRY(al, hipEventCreateWithFlags(&al->up_fork, hipEventDisableTiming));
      HIP_TRY(al, hipEventCreateWithFlags(&al->up_join, hipEventDisableTiming));
    }
    HIP_TRY(al, hipEventRecord(al->up_fork, stream));
    HIP_TRY(al, hipStreamWaitEvent(al->up_stream, al->up_fork, 0));
  }
  hipStream_t const main_stream = stream;
  int64_t src_lo = 0, src_hi = 0;
  if (!pieces.empty()) { src_lo = p_off[pieces.front().lo]; src_hi = t_off[pieces.back().hi - 1] + (in2bit ? (t_len[pieces.back().hi - 1] + 3) / 4 : t_len[pieces.back().hi - 1]); }
  const bool bind = upload_binds(al, seqs + std::min(src_lo, src_hi), (size_t)std::llabs(src_hi - src_lo));
  auto worker = [&](bool spawned) {
    bind_upload_worker(al, bind && spawned);   // (never the caller's own thread: ADVICE r05)
    (void)hipSetDevice(device);
    for (;;) {
      const long i = next.fetch_add(1);
      if (i >= np || failed.load()) return;
      const PackPiece& pc = pieces[(size_t)i];
      hipStream_t stream = (two_up && (i & 1)) ? al->up_stream : main_stream;   // (shadows the parameter: this piece's stream)
      const int sl = (int)(i % nslots);
      if (i >= nslots) while (issued[(size_t)sl].load(std::memory_order_acquire) != i - nslots) { if (failed.load()) return; std::this_thread::yield(); }
      if (al->pin_ev_recorded[(size_t)sl] && hipEventSynchronize(al->pin_ev[(size_t)sl]) != hipSuccess) failed.store(1);
      uint32_t* words = reinterpret_cast<uint32_t*>(al->pin_slot[(size_t)sl]);
      const size_t meta_at = ((size_t)pc.nwords * 4 + 15) & ~(size_t)15;
      WfaPairMeta* meta = reinterpret_cast<WfaPairMeta*>(al->pin_slot[(size_t)sl] + meta_at);
      uint32_t* const l16 = reinterpret_cast<uint32_t*>(al->pin_slot[(size_t)sl] + meta_at);
      uint64_t w = pc.wlo;
      for (int64_t q = pc.lo; q < pc.hi; ++q) {
        const int pl = p_len[q], tl = t_len[q];
        if (p_off[q] < 0 || t_off[q] < 0) { failed.store(2); return; }
        const uint32_t pw = (uint32_t)w;
        bool bd = false;
        if (in2bit) wfa::host_repack2_seq(seqs + p_off[q], pl, words + (w - pc.wlo));
        else bd = wfa::host_pack_seq(seqs + p_off[q], pl, words + (w - pc.wlo), -1);
        w += (uint64_t)((pl + 15) >> 4);
        const uint32_t tw = (uint32_t)w;
        if (in2bit) wfa::host_repack2_seq(seqs + t_off[q], tl, words + (w - pc.wlo));
        else bd |= wfa::host_pack_seq(seqs + t_off[q], tl, words + (w - pc.wlo), -1);
        w += (uint64_t)((tl + 15) >> 4);
        if (d_len16) l16[q - pc.lo] = (uint32_t)pl | ((uint32_t)tl << 16);
        else { WfaPairMeta& m = meta[q - pc.lo]; m.plen = pl; m.tlen = tl; m.p_woff = pw; m.t_woff = tw; }
        if (bd) bad[(size_t)i].push_back((uint32_t)q);
      }
      bool ok = true;
      if (pc.nwords) ok = hipMemcpyAsync(b->d_words + pc.wlo, words, (size_t)pc.nwords * 4, hipMemcpyHostToDevice, stream) == hipSuccess;
      if (d_len16) ok = ok && hipMemcpyAsync(d_len16 + pc.lo, l16, (size_t)(pc.hi - pc.lo) * sizeof(uint32_t), hipMemcpyHostToDevice, stream) == hipSuccess;
      else ok = ok && hipMemcpyAsync(b->d_meta + pc.lo, meta, (size_t)(pc.hi - pc.lo) * sizeof(WfaPairMeta), hipMemcpyHostToDevice, stream) == hipSuccess;
      ok = ok && hipEventRecord(al->pin_ev[(size_t)sl], stream) == hipSuccess;
      if (!ok) failed.store(1);
      al->pin_ev_recorded[(size_t)sl] = 1;
      issued[(size_t)sl].store(i, std::memory_order_release);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, true);
  worker(false);
  for (auto& x : th) x.join();
  if (two_up) {
    HIP_TRY(al, hipEventRecord(al->up_join, al->up_stream));
    HIP_TRY(al, hipStreamWaitEvent(stream, al->up_join, 0));
  }
  if (failed.load() == 2) { al->err = "negative length or offset"; return WFA_HIP_EINVAL; }
  if (failed.load()) { al->err = "pipelined upload failed"; return WFA_HIP_EDEVICE; }
  for (auto& v : bad) flagged->insert(flagged->end(), v.begin(), v.end());
  return WFA_HIP_OK;
}

static int pilot_first_width(wfa_hip_aligner* al, wfa_hip_batch* b, hipStream_t stream);
static int pilot_lane_heur(wfa_hip_aligner* al, wfa_hip_batch* b, hipStream_t stream);
static int pilot_band(wfa_hip_aligner* al, wfa_hip_batch* b, hipStream_t stream);

// in2bit: `seqs` holds 2-bit reads in the reference's packed form (wavefront_sequences.c:102-139: four bases per byte, base j
// of a byte in bits 2j..2j+1, A 0 / C 1 / G 2 / T 3), a sequence of len bases = (len + 3) / 4 bytes at its BYTE offset
static int batch_build(wfa_hip_aligner* al, wfa_hip_batch* b, int64_t n, const uint8_t* seqs,
                       const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len, bool in2bit = false) {
  if (al->mb_h && __atomic_load_n(&al->mb_h->alive, __ATOMIC_ACQUIRE) != 0) mailbox_quit(al);   // (the resident one-pair kernel: batches take the device)
  b->cfg = al->cfg; b->dcfg = al->dcfg; b->ncomp = al->ncomp; b->gcfg = al->gcfg; b->gncomp = al->gncomp;
  // Round 5: a wildcard letter outside ACGT cannot occur in a pair whose letters are all ACGT — such pairs take the 2-bit kernels as if
  // no wildcard were set (same result: nothing in them matches by wildcard), only the pairs holding other letters are aligned on their
  // bytes with the wildcard rule (150 bp, clean reads: 44 M -> the 2-bit rates).  A wildcard that IS one of ACGT keeps every pair on bytes.
  {
    const int wc = b->cfg.wildcard;
    if (wc >= 0 && !(wc == 'A' || wc == 'C' || wc == 'G' || wc == 'T' || wc == 'a' || wc == 'c' || wc == 'g' || wc == 't')) { b->wild = wc; b->dcfg.wildcard = -1; b->gcfg.wildcard = -1; }
  }
  const wfa_hip_config_t& c = b->cfg;
  const bool timing = al->knobs.set[K_TIMING];
  double t0 = now_ms();
  b->al = al;
  b->n = n;
  if (c.scope == WFA_SCOPE_FULL) {  // needed later to lay out the op-string regions
    b->h_plen.assign(p_len, p_len + n);
    b->h_tlen.assign(t_len, t_len + n);
  }
  // per-pair metadata on several host threads: pass 1 validates and sums each slice of the batch, pass 2 writes the word
  // offsets from the slice prefix.  Large batches (pipelined upload): the slices are the pieces of the upload ring and pass 2
  // writes each piece straight into its pinned slot — no host copy of the metadata exists
  // (round 4: also few pairs of long reads — C5's 100 kb reads are 200 KB of ASCII per pair: a plain copy from pageable memory moved
  // them at 3.6 GB/s, the pinned ring with host-side packing at the link's rate)
  bool many_bases = false;
  if (n < 262144 && n >= 64) {
    int64_t bases = 0;
    for (int64_t i = 0; i < n; ++i) bases += (int64_t)std::max(p_len[i], 0) + std::max(t_len[i], 0);
    many_bases = bases >= ((int64_t)32 << 20);
  }
  const bool pipelined = (n >= 262144 || many_bases) && knob(al, K_NO_PIPE, 0) == 0;
  const int64_t piece_pairs = (int64_t)(staged_slot_bytes(al) / sizeof(WfaPairMeta));
  // (pipelined: a piece of the metadata upload = `sub` parts of pass 1, so that a large team shares pass 1 evenly)
  const int sub = pipelined ? 8 : 1;
  const int64_t part_pairs = piece_pairs / sub;   // (a multiple of 64: the slot size is whole megabytes)
  const int nthr = pipelined ? (int)((n + part_pairs - 1) / part_pairs)
                             : (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, std::thread::hardware_concurrency()), n / 65536));
  auto part_lo = [&](int t) -> int64_t { return pipelined ? std::min<int64_t>(n, (int64_t)t * part_pairs) : n * t / nthr; };
  std::unique_ptr<WfaPairMeta[]> meta(pipelined ? nullptr : new WfaPairMeta[(size_t)std::max<int64_t>(n, 1)]);  // not zero-filled: first touched in parallel below
  struct Part { uint64_t words = 0; int64_t packed = 0, ops = 0, blob_end = 0; int max_width = 0, max_len = 0, err = 0; };
  std::vector<Part> parts((size_t)nthr);
  // host-packed upload: large batches without a wildcard (a wildcard aligns every pair on its bytes).  Pass 1 then also
  // sums the words of every block of 64 pairs (the parts start on block boundaries), the pieces of the upload are cut
  // from those below
  // (round 6: 2-bit input too — its bytes are re-based to whole words and re-coded by the same workers: wfa::host_repack2_seq)
  bool host_pack = pipelined && c.wildcard < 0 && knob(al, K_HOST_PACK, 1) != 0;
  const bool light = host_pack;   // (pass 1 reads the lengths only; see pass1_light)
  std::vector<uint32_t> blk_words(host_pack ? (size_t)((n + 63) >> 6) : 0, 0u);
  auto pass1 = [&](int t) {
    Part& pt = parts[(size_t)t];
    const int64_t lo = part_lo(t), hi = part_lo(t + 1);
    for (int64_t i = lo; i < hi; ++i) {
      const int pl = p_len[i], tl = t_len[i];
      // (host-packed upload: the offsets are first read, and checked, by the pack workers; pass 1 reads the lengths only)
      if (pl < 0 || tl < 0 || (!light && (p_off[i] < 0 || t_off[i] < 0))) { pt.err = 1; return; }
      if ((int64_t)pl + tl > (int64_t)INT_MAX / 2 - 8) { pt.err = 2; return; }
      // wavefront_align.c:86-102: the reference exit(1)s here
      if (c.span == WFA_SPAN_ENDSFREE &&
          (c.pattern_begin_free > pl || c.pattern_end_free > pl || c.text_begin_free > tl || c.text_end_free > tl)) { pt.err = 3; return; }
      pt.words += (uint64_t)((pl + 15) >> 4) + (uint64_t)((tl + 15) >> 4);
      if (host_pack) blk_words[(size_t)(i >> 6)] += (uint32_t)((pl + 15) >> 4) + (uint32_t)((tl + 15) >> 4);
      pt.max_width = std::max(pt.max_width, pl + tl + 3);
      pt.max_len = std::max(pt.max_len, std::max(pl, tl));
      pt.packed += (int64_t)((pl + 3) >> 2) + ((tl + 3) >> 2);
      pt.ops += (int64_t)pl + tl;
      if (!light) pt.blob_end = in2bit ? std::max(pt.blob_end, std::max(p_off[i] + ((pl + 3) >> 2), t_off[i] + ((tl + 3) >> 2)))
                                       : std::max(pt.blob_end, std::max(p_off[i] + pl, t_off[i] + tl));
    }
  };
  // pass 1 of the host-packed form: lengths only, no early exits — a loop the compiler vectorises (10 M pairs: 3.3 -> ~0.5 ms
  // on 32 threads); negative lengths / free ends larger than a sequence are found from the OR / minima afterwards
  auto pass1_light = [&](int t) {
    Part& pt = parts[(size_t)t];
    const int64_t lo = part_lo(t), hi = part_lo(t + 1);
    int32_t neg = 0, minp = INT_MAX, mint = INT_MAX, mxl = 0;
    uint32_t mxw = 0;
    uint64_t words = 0, packed = 0, ops = 0;
    for (int64_t b0 = lo; b0 < hi; b0 += 64) {
      const int64_t b1 = std::min<int64_t>(hi, b0 + 64);
      uint32_t bw = 0, bp = 0, bo = 0;
      for (int64_t i = b0; i < b1; ++i) {
        const int32_t pl = p_len[i], tl = t_len[i];
        neg |= pl | tl;
        minp = std::min(minp, pl); mint = std::min(mint, tl);
        mxl = std::max(mxl, std::max(pl, tl));
        mxw = std::max(mxw, (uint32_t)pl + (uint32_t)tl);
        bw += ((uint32_t)(pl + 15) >> 4) + ((uint32_t)(tl + 15) >> 4);
        bp += ((uint32_t)(pl + 3) >> 2) + ((uint32_t)(tl + 3) >> 2);
        bo += (uint32_t)pl + (uint32_t)tl;
      }
      // (64 pairs: the 32-bit sums cannot overflow before the length check below fails)
      blk_words[(size_t)(b0 >> 6)] = bw;
      words += bw; packed += bp; ops += bo;
    }
    if (neg < 0) { pt.err = 1; return; }
    if (mxw > (uint32_t)(INT_MAX / 2 - 8)) { pt.err = 2; return; }
    if (mxl > (1 << 24)) { pt.err = 4; return; }   // (sequences this long: the plain form and its 64-bit sums)
    if (hi > lo && c.span == WFA_SPAN_ENDSFREE &&
        (std::max(c.pattern_begin_free, c.pattern_end_free) > minp || std::max(c.text_begin_free, c.text_end_free) > mint)) { pt.err = 3; return; }
    pt.words = words; pt.packed = (int64_t)packed; pt.ops = (int64_t)ops;
    pt.max_width = (int)mxw + 3; pt.max_len = mxl;
  };
  auto pass_offsets = [&](int t) {   // what a light pass 1 left out (the host-packed form turned out not to fit)
    Part& pt = parts[(size_t)t];
    for (int64_t i = part_lo(t), hi = part_lo(t + 1); i < hi; ++i) {
      if (p_off[i] < 0 || t_off[i] < 0) { pt.err = 1; return; }
      pt.blob_end = std::max(pt.blob_end, std::max(p_off[i] + (int64_t)p_len[i], t_off[i] + (int64_t)t_len[i]));
    }
  };
  auto run_threads = [&](auto&& fn) {
    if (nthr == 1) { fn(0); return; }
    // (a bounded team: the parts are claimed from a counter)
    const int team = std::min(nthr, std::max(1, std::min(pipelined ? 32 : 16, (int)std::thread::hardware_concurrency() / (pipelined ? 2 : 1))));
    std::atomic<int> nextp(0);
    auto loop = [&]() { for (int t = nextp.fetch_add(1); t < nthr; t = nextp.fetch_add(1)) fn(t); };
    std::vector<std::thread> th;
    for (int t = 1; t < team; ++t) th.emplace_back(loop);
    loop();
    for (auto& x : th) x.join();
  };
  if (light) {
    run_threads(pass1_light);
    bool too_long = false;
    for (const Part& pt : parts) too_long |= (pt.err == 4);
    if (too_long) {
      host_pack = false;
      for (Part& pt : parts) pt = Part();
      run_threads(pass1);
      run_threads(pass_offsets);
    }
  } else run_threads(pass1);
  uint64_t woff = 0;
  int64_t blob_end = 0;
  std::vector<uint64_t> wbase((size_t)nthr);
  for (int t = 0; t < nthr; ++t) {
    const Part& pt = parts[(size_t)t];
    if (pt.err == 1) { al->err = "negative length or offset"; return WFA_HIP_EINVAL; }
    if (pt.err == 2) { al->err = "sequence too long"; return WFA_HIP_EINVAL; }
    if (pt.err == 3) { al->err = "Ends-free parameters must be not larger than the sequences"; return WFA_HIP_EINVAL; }
    wbase[(size_t)t] = woff; woff += pt.words;
    b->max_width = std::max(b->max_width, pt.max_width); b->max_len = std::max(b->max_len, pt.max_len);
    b->packed_bytes += pt.packed; b->ops_bytes += pt.ops; blob_end = std::max(blob_end, pt.blob_end);
  }
  if (woff > 0xFFFFFFF0ull) { al->err = "batch too large: more than 2^32 packed words (split the batch)"; return WFA_HIP_EINVAL; }
  std::vector<PackPiece> pack_pieces;
  // the metadata of a host-packed upload as 4 B per pair (16-bit lengths; the word offsets are rebuilt on the device) when no sequence
  // is longer than 65 535 bases (WFA_HIP_LEN16=0: the 16 B records)
  const bool len16 = host_pack && b->max_len < 65536 && knob(al, K_LEN16, 1) != 0;
  const size_t meta_wire = len16 ? sizeof(uint32_t) : sizeof(WfaPairMeta);
  if (host_pack) {
    const size_t slot = staged_slot_bytes(al);
    PackPiece cur{0, 0, 0, 0};
    const int64_t nblk = (n + 63) >> 6;
    for (int64_t bi = 0; bi < nblk && host_pack; ++bi) {
      const int64_t blo = bi << 6, bhi = std::min<int64_t>(n, blo + 64);
      const uint64_t bw = blk_words[(size_t)bi];
      if ((size_t)bw * 4 + 16 + 64 * sizeof(WfaPairMeta) > slot) host_pack = false;   // (very long reads: the plain form)
      if (cur.hi > cur.lo && (size_t)(cur.nwords + bw) * 4 + 16 + (size_t)(bhi - cur.lo) * meta_wire > slot) {
        pack_pieces.push_back(cur);
        cur = PackPiece{blo, blo, cur.wlo + cur.nwords, 0};
      }
      cur.hi = bhi; cur.nwords += bw;
    }
    if (cur.hi > cur.lo) pack_pieces.push_back(cur);
    if (!host_pack) {
      run_threads(pass_offsets);
      for (int t = 0; t < nthr; ++t) {
        if (parts[(size_t)t].err == 1) { al->err = "negative length or offset"; return WFA_HIP_EINVAL; }
        blob_end = std::max(blob_end, parts[(size_t)t].blob_end);
      }
    }
  }
  auto pass2_part = [&](int t, WfaPairMeta* out /* element 0 = pair part_lo(t) */, int nparts = 1) {
    uint64_t w = wbase[(size_t)t];
    const int64_t lo = part_lo(t), hi = part_lo(std::min(t + nparts, nthr));
    for (int64_t i = lo; i < hi; ++i) {
      const int pl = p_len[i], tl = t_len[i];
      WfaPairMeta& m = out[i - lo];
      m.p_woff = (uint32_t)w; w += (uint64_t)((pl + 15) >> 4);
      m.t_woff = (uint32_t)w; w += (uint64_t)((tl + 15) >> 4);
      m.plen = pl; m.tlen = tl;
    }
  };
  if (!pipelined) run_threads([&](int t) { pass2_part(t, meta.get() + part_lo(t)); });
  if (timing) { fprintf(stderr, "[wfa_hip] meta pass %.3f ms\n", now_ms() - t0); t0 = now_ms(); }
  const bool full = (c.scope == WFA_SCOPE_FULL);
  const size_t nn = (size_t)std::max<int64_t>(n, 1);
  if (!host_pack) HIP_TRY(al, pool_alloc(al, (void**)&b->d_bytes, (size_t)blob_end + 64));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_pboff, nn * sizeof(int64_t)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_tboff, nn * sizeof(int64_t)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_meta, nn * sizeof(WfaPairMeta)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_words, ((size_t)woff + 4) * sizeof(uint32_t)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_flags, nn));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_score, nn * sizeof(int32_t)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_status, nn * sizeof(int32_t)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_fb_list2[0], nn * sizeof(uint32_t)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_fb_list2[1], nn * sizeof(uint32_t)));
  HIP_TRY(al, pool_alloc(al, (void**)&b->d_counters, WFA_COUNTER_WORDS * sizeof(uint32_t)));
  HIP_TRY(al, hipMemsetAsync(b->d_counters, 0, WFA_COUNTER_WORDS * sizeof(uint32_t), al->stream));
  HIP_TRY(al, hipMemsetAsync(b->d_flags, 0, nn, al->stream));
  HIP_TRY(al, hipMemsetAsync(b->d_words + woff, 0, 4 * sizeof(uint32_t), al->stream));
  if (full) {
    std::vector<int64_t>& coff = b->h_coff;  // (a member: its upload may outlive this function for small batches)
    coff.assign((size_t)n + 1, 0);
    coff[0] = 0;
    for (int64_t i = 0; i < n; ++i) coff[i + 1] = coff[i] + p_len[i] + t_len[i];
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_ops, (size_t)std::max<int64_t>(b->ops_bytes, 1)));
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_cigar_off, ((size_t)n + 1) * sizeof(int64_t)));
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_cigar_begin, nn * sizeof(int64_t)));
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_cigar_len, nn * sizeof(int32_t)));
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_ovf_list[0], nn * sizeof(uint32_t)));
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_ovf_list[1], nn * sizeof(uint32_t)));
    HIP_TRY(al, hipMemcpyAsync(b->d_cigar_off, coff.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, al->stream));
    if (n > 256) HIP_TRY(al, hipStreamSynchronize(al->stream)); else b->uploads_pending = true;
  }
  if (timing) { fprintf(stderr, "[wfa_hip] mallocs %.3f ms\n", now_ms() - t0); t0 = now_ms(); }
  if (n > 0 && host_pack) {
    std::vector<uint32_t> flagged;
    const bool use16 = len16 && host_pack;
    if (use16) {
      HIP_TRY(al, pool_alloc(al, (void**)&b->d_len16, nn * sizeof(uint32_t)));
      HIP_TRY(al, pool_alloc(al, (void**)&b->d_pieces, pack_pieces.size() * sizeof(wfa::WfaPieceDesc)));
    }
    const int urc = staged_pack_upload(al, b, pack_pieces, seqs, p_off, p_len, t_off, t_len, al->stream, &flagged, in2bit, use16 ? b->d_len16 : nullptr);
    if (urc != WFA_HIP_OK) return urc;
    if (use16) {
      b->h_pieces.resize(pack_pieces.size());
      for (size_t q = 0; q < pack_pieces.size(); ++q) b->h_pieces[q] = wfa::WfaPieceDesc{(long long)pack_pieces[q].lo, (long long)pack_pieces[q].hi, (unsigned long long)pack_pieces[q].wlo};
      HIP_TRY(al, hipMemcpyAsync(b->d_pieces, b->h_pieces.data(), b->h_pieces.size() * sizeof(wfa::WfaPieceDesc), hipMemcpyHostToDevice, al->stream));
      hipLaunchKernelGGL(wfa::wfa_meta_from_len16_kernel, dim3((unsigned)pack_pieces.size()), dim3(256), 0, al->stream, b->d_pieces, b->d_len16, b->d_meta);
      HIP_TRY(al, hipGetLastError());
    }
    if (timing) { fprintf(stderr, "[wfa_hip] host pack + H2D enqueue %.3f ms (%.2f GB of ASCII, %.2f GB sent)\n", now_ms() - t0, b->ops_bytes / 1e9,
                          (woff * 4.0 + n * 16.0) / 1e9); t0 = now_ms(); }
    b->n_bytes = (uint32_t)flagged.size();
    b->n_packed = (uint32_t)n;
    if (!flagged.empty()) {
      // the few pairs with other letters: their bytes in a compact blob, byte offsets and flags scattered to their slots
      const size_t nb = flagged.size();
      std::vector<int64_t> fpo(nb), fto(nb);
      size_t blob = 0;
      for (size_t j = 0; j < nb; ++j) { fpo[j] = (int64_t)blob; blob += (size_t)p_len[flagged[j]]; fto[j] = (int64_t)blob; blob += (size_t)t_len[flagged[j]]; }
      std::vector<uint8_t> hb(blob + 1);
      for (size_t j = 0; j < nb; ++j) {
        memcpy(hb.data() + fpo[j], seqs + p_off[flagged[j]], (size_t)p_len[flagged[j]]);
        memcpy(hb.data() + fto[j], seqs + t_off[flagged[j]], (size_t)t_len[flagged[j]]);
      }
      int64_t *d_fpo = nullptr, *d_fto = nullptr;
      HIP_TRY(al, pool_alloc(al, (void**)&b->d_bytes, blob + 64));
      HIP_TRY(al, pool_alloc(al, (void**)&b->d_list_bytes, nb * sizeof(uint32_t)));
      HIP_TRY(al, pool_alloc(al, (void**)&d_fpo, nb * sizeof(int64_t)));
      HIP_TRY(al, pool_alloc(al, (void**)&d_fto, nb * sizeof(int64_t)));
      HIP_TRY(al, hipMemcpyAsync(b->d_bytes, hb.data(), blob, hipMemcpyHostToDevice, al->stream));
      HIP_TRY(al, hipMemcpyAsync(b->d_list_bytes, flagged.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, al->stream));
      HIP_TRY(al, hipMemcpyAsync(d_fpo, fpo.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, al->stream));
      HIP_TRY(al, hipMemcpyAsync(d_fto, fto.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, al->stream));
      hipLaunchKernelGGL(wfa::wfa_flag_scatter_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, al->stream,
                         b->d_list_bytes, d_fpo, d_fto, (uint32_t)nb, b->d_pboff, b->d_tboff, b->d_flags);
      HIP_TRY(al, hipGetLastError());
      // the list of the 2-bit pairs = all pairs but the flagged ones (ascending)
      std::vector<uint32_t> lp;
      lp.reserve((size_t)n - nb);
      size_t fj = 0;
      for (int64_t i = 0; i < n; ++i) { if (fj < nb && flagged[fj] == (uint32_t)i) { ++fj; continue; } lp.push_back((uint32_t)i); }
      b->n_packed = (uint32_t)lp.size();
      if (!lp.empty()) {
        HIP_TRY(al, pool_alloc(al, (void**)&b->d_list_packed, lp.size() * sizeof(uint32_t)));
        HIP_TRY(al, hipMemcpyAsync(b->d_list_packed, lp.data(), lp.size() * sizeof(uint32_t), hipMemcpyHostToDevice, al->stream));
      }
      HIP_TRY(al, hipStreamSynchronize(al->stream));   // (host vectors above go out of scope)
      pool_release(al, d_fpo); pool_release(al, d_fto);
    }
  } else if (n > 0) {
    if (pipelined) {
      // host threads -> pinned ring -> DMA; the metadata pieces are computed into their slots, the caller's arrays copied
      std::vector<UploadJob> jobs;
      jobs.push_back({b->d_meta, nullptr, (size_t)n * sizeof(WfaPairMeta),
                      [&](uint8_t* out, size_t off, size_t) { pass2_part((int)(off / staged_slot_bytes(al)) * sub, reinterpret_cast<WfaPairMeta*>(out), sub); }});
      jobs.push_back({b->d_bytes, seqs, (size_t)blob_end, nullptr});
      jobs.push_back({b->d_pboff, p_off, (size_t)n * sizeof(int64_t), nullptr});
      jobs.push_back({b->d_tboff, t_off, (size_t)n * sizeof(int64_t), nullptr});
      const int urc = staged_upload(al, jobs, al->stream);
      if (urc != WFA_HIP_OK) return urc;
    } else {
      HIP_TRY(al, hipMemcpyAsync(b->d_bytes, seqs, (size_t)blob_end, hipMemcpyHostToDevice, al->stream));
      HIP_TRY(al, hipMemcpyAsync(b->d_pboff, p_off, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, al->stream));
      HIP_TRY(al, hipMemcpyAsync(b->d_tboff, t_off, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, al->stream));
      HIP_TRY(al, hipMemcpyAsync(b->d_meta, meta.get(), (size_t)n * sizeof(WfaPairMeta), hipMemcpyHostToDevice, al->stream));
    }
    if (in2bit) {
      // 2-bit input: re-based to whole words and re-coded by one small kernel; no letter can be outside ACGT
      const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)al->cu_count * 64));
      hipLaunchKernelGGL(wfa::wfa_repack2_kernel, dim3(grid), dim3(256), 0, al->stream, b->d_bytes, b->d_pboff, b->d_tboff, b->d_meta, n, b->d_words);
      HIP_TRY(al, hipGetLastError());
      b->n_bytes = 0; b->n_packed = (uint32_t)n;
      if (n <= 256) b->uploads_pending = true;
      else {   // (plain copies: the caller's arrays are still being read; either way nothing reads the 2-bit blob after the repack kernel —
               // it used to stay in HBM for the batch's lifetime on the pipelined path)
        HIP_TRY(al, hipStreamSynchronize(al->stream));
        pool_release(al, b->d_bytes); b->d_bytes = nullptr;
      }
      b->h_meta = std::move(meta);
      { const int prc = pilot_first_width(al, b, al->stream); if (prc != WFA_HIP_OK) return prc; }
      { const int prc = pilot_lane_heur(al, b, al->stream); if (prc != WFA_HIP_OK) return prc; }
      { const int prc = pilot_band(al, b, al->stream); if (prc != WFA_HIP_OK) return prc; }
      HIP_TRY(al, hipEventCreateWithFlags(&b->upload_event, hipEventDisableTiming));
      HIP_TRY(al, hipEventRecord(b->upload_event, al->stream));
      return WFA_HIP_OK;
    }
    const int threads = 256;
    // lanes per pair: the words of the longest pair, rounded up to a power of two (at most a whole wave)
    int log2slots = 1;
    while (log2slots < 6 && (1 << log2slots) < 2 * ((b->max_len + 15) >> 4)) ++log2slots;
    const int64_t group = 64 >> log2slots;
    const int64_t want = ((n + group - 1) / group + 3) / 4;  // 4 waves per workgroup
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)al->cu_count * 64));
    hipLaunchKernelGGL(wfa::wfa_pack_kernel, dim3(grid), dim3(threads), 0, al->stream,
                       b->d_bytes, b->d_pboff, b->d_tboff, b->d_meta, n, b->d_words, b->d_flags, log2slots, b->d_counters + 15);
    HIP_TRY(al, hipGetLastError());
    std::vector<uint8_t> flags;
    if (n <= 256) {
      flags.assign((size_t)n, 0);
      // a handful of pairs: look for letters outside ACGT on the host, the round trip costs more than the scan
      for (int64_t i = 0; i < n; ++i) {
        uint8_t bad = 0;
        const uint8_t* ps = seqs + p_off[i]; const uint8_t* ts = seqs + t_off[i];
        for (int32_t j = 0; j < p_len[i]; ++j) { const uint8_t ch = ps[j]; bad |= !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
        for (int32_t j = 0; j < t_len[i]; ++j) { const uint8_t ch = ts[j]; bad |= !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
        flags[(size_t)i] = bad;
      }
      b->uploads_pending = true;  // the caller's arrays are still being read: see wfa_hip_batch_create
    } else {
      // one word first: the per-pair flags (n bytes) are fetched only if some pair holds a letter outside ACGT
      uint32_t any_flag = 0;
      HIP_TRY(al, hipMemcpyAsync(&any_flag, b->d_counters + 15, sizeof(uint32_t), hipMemcpyDeviceToHost, al->stream));
      HIP_TRY(al, hipStreamSynchronize(al->stream));
      if (any_flag || (c.wildcard >= 0 && b->wild < 0)) {
        flags.assign((size_t)n, 0);
        HIP_TRY(al, hipMemcpy(flags.data(), b->d_flags, (size_t)n, hipMemcpyDeviceToHost));
      }
    }
    if (timing) { fprintf(stderr, "[wfa_hip] H2D + pack + flags D2H %.3f ms (%.2f GB)\n", now_ms() - t0, blob_end / 1e9); t0 = now_ms(); }
    // split into the 2-bit and the 8-bit work lists (wildcard matching needs the bytes)
    std::vector<uint32_t> lp, lb;
    if (c.wildcard >= 0 && b->wild < 0) {   // (a wildcard among ACGT: every pair on its bytes)
      lb.resize((size_t)n);
      for (int64_t i = 0; i < n; ++i) lb[i] = (uint32_t)i;
    } else {
      int64_t nbad = 0;
      for (size_t i = 0; i < flags.size(); ++i) nbad += flags[i];
      if (nbad) {
        lp.reserve((size_t)(n - nbad)); lb.reserve((size_t)nbad);
        for (int64_t i = 0; i < n; ++i) (flags[i] ? lb : lp).push_back((uint32_t)i);
      }
    }
    b->n_bytes = (uint32_t)lb.size();
    b->n_packed = (uint32_t)(lb.empty() ? n : lp.size());
    if (!lb.empty()) {
      HIP_TRY(al, pool_alloc(al, (void**)&b->d_list_bytes, lb.size() * sizeof(uint32_t)));
      HIP_TRY(al, hipMemcpy(b->d_list_bytes, lb.data(), lb.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      if (!lp.empty()) {
        HIP_TRY(al, pool_alloc(al, (void**)&b->d_list_packed, lp.size() * sizeof(uint32_t)));
        HIP_TRY(al, hipMemcpy(b->d_list_packed, lp.data(), lp.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      }
    } else {
      // every pair is pure ACGT: the ASCII blob is no longer needed
      pool_release(al, b->d_bytes); b->d_bytes = nullptr;
    }
    if (timing) { fprintf(stderr, "[wfa_hip] work lists + free %.3f ms\n", now_ms() - t0); t0 = now_ms(); }
  }
  b->h_meta = std::move(meta);
  { const int prc = pilot_first_width(al, b, al->stream); if (prc != WFA_HIP_OK) return prc; }
  { const int prc = pilot_lane_heur(al, b, al->stream); if (prc != WFA_HIP_OK) return prc; }
  { const int prc = pilot_band(al, b, al->stream); if (prc != WFA_HIP_OK) return prc; }
  HIP_TRY(al, hipEventCreateWithFlags(&b->upload_event, hipEventDisableTiming));
  HIP_TRY(al, hipEventRecord(b->upload_event, al->stream));
  return WFA_HIP_OK;
}

// The narrowest band that keeps most pairs is the cheapest first stage of the short-read cascade; it depends on the divergence
// of the batch (16 lanes up to ~3 %, 32 up to ~6 %, 64 up to ~10 %).  A pilot on 8192 pairs sampled at a fixed stride across
// the batch (score-only kernels, one-round form: a name of its own in a profile) decides once per batch, when the batch is
// created: b->stage_pick = 16 / 32 / 64, or 128 = none of them.  (Up to three small launches, each waited for: it runs where
// the upload is waited for anyway, so that wfa_hip_batch_run only enqueues.)
// score_mode (csrc/wfa_common.hpp): the kernels leave -s; completed pairs get the score of the original configuration
__global__ void __launch_bounds__(256) wfa_score_translate_kernel(int32_t* __restrict__ score, const int32_t* __restrict__ status,
                                                                  const WfaPairMeta* __restrict__ meta, long long n, int mode, int sw_match) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || status[i] != 0) return;
  const int raw = score[i];   // -s
  if (mode == 2) score[i] = -raw;
  else score[i] = (int)(((long long)sw_match * ((long long)meta[i].plen + meta[i].tlen) + raw) / 2);
}

// ... of the listed pairs only (full scope: the pairs whose arena overflowed are re-run after the stream's translation pass)
__global__ void __launch_bounds__(256) wfa_score_translate_list_kernel(const uint32_t* __restrict__ list, uint32_t count, int32_t* __restrict__ score,
                                                                       const int32_t* __restrict__ status, const WfaPairMeta* __restrict__ meta,
                                                                       int mode, int sw_match) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= count) return;
  const uint32_t i = list[j];
  if (status[i] != 0) return;
  const int raw = score[i];
  if (mode == 2) score[i] = -raw;
  else score[i] = (int)(((long long)sw_match * ((long long)meta[i].plen + meta[i].tlen) + raw) / 2);
}

__global__ void __launch_bounds__(256) wfa_pilot_sample_kernel(const uint32_t* __restrict__ list, uint32_t stride, uint32_t np, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < np) out[i] = list ? list[i * stride] : i * stride;
}

// A pilot's launch failed.  If the run-time path failed under it (wfa::rtc_failure_count moved since `failures`), the batch and its
// aligner go on without run-time shapes and the pilot's stage is left out (WFA_HIP_OK); any other failure is the device's.
static int pilot_launch_failed(wfa_hip_aligner* al, wfa_hip_batch* b, unsigned failures) {
  if (b->dcfg.rtc && wfa::rtc_failure_count() != failures) {
    (void)hipGetLastError();
    b->dcfg.rtc = 0; al->dcfg.rtc = 0;
    al->rtc_note = std::string("run-time kernels switched off for this aligner: ") + wfa::rtc_last_error();
    b->stage_pick = 128; b->laneh_pick = 2; b->segh_pick = 2; if (b->band_pick == 0) b->band_pick = 1;
    return WFA_HIP_OK;
  }
  al->err = "pilot launch failed";
  return WFA_HIP_EDEVICE;
}

static int pilot_first_width(wfa_hip_aligner* al, wfa_hip_batch* b, hipStream_t stream) {
  const unsigned rtc_failures = wfa::rtc_failure_count();
  const bool full = (b->cfg.scope == WFA_SCOPE_FULL);
  if (b->stage_pick != 0 || b->n_packed < 65536u) return WFA_HIP_OK;
  if (!wfa::seg_supported(b->dcfg, b->ncomp, false) || b->max_len > WFA_FAST_MAX_LEN || knob(al, K_NO_FAST, 0) != 0) return WFA_HIP_OK;
  if (full ? (knob(al, K_NO_SEGFULL, 0) != 0 || al->knobs.set[K_SEGFULL_STAGES]) : !al->knobs.fast_stages.empty()) return WFA_HIP_OK;
  const uint32_t np = 8192u, stride = b->n_packed / np;
  uint32_t* plist = b->d_fb_list2[0];
  uint32_t* psample = b->d_fb_list2[1];
  uint32_t* pcount = b->d_counters + 4;
  hipLaunchKernelGGL(wfa_pilot_sample_kernel, dim3((np + 255u) / 256u), dim3(256), 0, stream, b->d_list_packed, stride, np, psample);
  HIP_TRY(al, hipGetLastError());
  b->stage_pick = 128;
  for (int w = 16; w <= 64; w *= 2) {
    HIP_TRY(al, hipMemsetAsync(pcount, 0, sizeof(uint32_t), stream));
    if (wfa::launch_seg(b->dcfg, al->cu_count, knob(al, K_FAST_WAVES_PER_CU, 256), stream, b->d_words, b->d_meta, psample, nullptr, np, b->d_score, b->d_status,
                        plist, pcount, w == 16 ? 2 : (w == 32 ? 4 : 5)) != 0) return pilot_launch_failed(al, b, rtc_failures);
    uint32_t handed = 0;
    HIP_TRY(al, hipMemcpyAsync(&handed, pcount, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(al, hipStreamSynchronize(stream));
    // (16 diagonals as the first stage pay while they hand on < ~8 % of the pairs score-only (5 %: 1.37 G against 1.21 G aln/s from 32
    // diagonals; 11 %: 0.87 against 1.10), < ~15 % with the CIGAR (11 %: 0.49 against 0.41; 19 %: 0.30 against 0.37): measured per
    // divergence, tools/probes/pilot_probe.py, profiles/r04_pilot_probe.txt.  At the 40 % of round 3 the 4 % batches ran 1.8x slower.)
    const uint32_t pct = (w == 16) ? (uint32_t)knob(al, K_PILOT_PCT, full ? 15 : 8) : 40u;
    if (knob(al, K_STAGE_TIMING, 0)) fprintf(stderr, "[wfa_hip] pilot: %d diagonals hand on %u of %u\n", w, handed, np);
    if (handed * 100u <= np * pct) { b->stage_pick = w; break; }
  }
  HIP_TRY(al, hipMemsetAsync(pcount, 0, sizeof(uint32_t), stream));
  return WFA_HIP_OK;
}

// Exact reads of 300 - 1 200 bases (no heuristic): the 256-diagonal register window is several times faster than the tiled rows for the
// pairs it can finish, and wasted work for those whose wavefront outgrows it (scores beyond ~250 steps: 1 kb at 5 %, 600 bp at 10 % —
// 91 / 99 % handed on, 17 - 20 % of the run).  A pilot on 4 096 sampled pairs decides once per batch (b->band_pick).
static int pilot_band(wfa_hip_aligner* al, wfa_hip_batch* b, hipStream_t stream) {
  const unsigned rtc_failures = wfa::rtc_failure_count();
  if (b->band_pick != 0 || b->n_packed < 32768u) return WFA_HIP_OK;
  const bool two = b->ncomp == 5;   // (gap-affine-2p: the 192- and the 256-diagonal stage, reads of up to 2 kb)
  if (b->dcfg.heuristic != WFA_HEUR_NONE || (b->ncomp != 3 && !two) || (two ? (b->max_len <= 100 || b->max_len > 2000) : (b->max_len <= 300 || b->max_len > 1200))) return WFA_HIP_OK;
  if (!wfa::band_supported(b->dcfg, b->ncomp) || knob(al, K_NO_BAND, 0) != 0 || knob(al, K_BAND_NCH, 0) != 0) return WFA_HIP_OK;
  const uint32_t np = 4096u, stride = b->n_packed / np;
  uint32_t* plist = b->d_fb_list2[0];
  uint32_t* psample = b->d_fb_list2[1];
  uint32_t* pcount = b->d_counters + 4;
  hipLaunchKernelGGL(wfa_pilot_sample_kernel, dim3((np + 255u) / 256u), dim3(256), 0, stream, b->d_list_packed, stride, np, psample);
  HIP_TRY(al, hipGetLastError());
  HIP_TRY(al, hipMemsetAsync(pcount, 0, sizeof(uint32_t), stream));
  wfa::BandArgs ba;
  memset(&ba, 0, sizeof(ba));
  ba.words = b->d_words; ba.meta = b->d_meta; ba.worklist = psample; ba.nwork = np;
  ba.score = b->d_score; ba.status = b->d_status; ba.fb_list = plist; ba.fb_count = pcount;
  ba.g = wfa::band_gcd(b->dcfg, two);
  ba.x = b->dcfg.x; ba.oe = b->dcfg.o1 + b->dcfg.e1; ba.e = b->dcfg.e1;
  if (two) { ba.oe2 = b->dcfg.o2 + b->dcfg.e2; ba.e2 = b->dcfg.e2; }
  ba.min_wf_len = b->dcfg.min_wf_len; ba.max_dist_thr = b->dcfg.max_dist_thr; ba.steps_between = b->dcfg.steps_between;
  ba.heur = b->dcfg.heuristic; ba.xdrop = b->dcfg.xdrop; ba.max_steps = b->dcfg.max_steps; ba.scope = b->dcfg.scope;
  const int words = ((b->max_len + 15) >> 4) + 4;
  ba.lds_words = words;
  ba.slim = knob(al, K_BAND_SLIM, 1);
  ba.h16 = 1;
  ba.ef = (b->dcfg.endsfree && (b->dcfg.pbf | b->dcfg.pef | b->dcfg.tbf | b->dcfg.tef)) ? 1 : 0;
  ba.pbf = b->dcfg.pbf; ba.pef = b->dcfg.pef; ba.tbf = b->dcfg.tbf; ba.tef = b->dcfg.tef;
  if (wfa::launch_band(ba, 4, false, false, true, (long long)np, stream) != 0) return pilot_launch_failed(al, b, rtc_failures);   // (the widest window: what it hands on, both stages hand on)
  uint32_t handed = 0;
  HIP_TRY(al, hipMemcpyAsync(&handed, pcount, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY(al, hipStreamSynchronize(stream));
  if (knob(al, K_STAGE_TIMING, 0)) fprintf(stderr, "[wfa_hip] pilot: 256-diagonal window hands on %u of %u\n", handed, np);
  b->band_pick = (handed * 2u >= np) ? 2 : 1;
  HIP_TRY(al, hipMemsetAsync(pcount, 0, sizeof(uint32_t), stream));
  return WFA_HIP_OK;
}

// The general score-only form of the lane kernel (wfa_lane_kernel<.., HEUR>) keeps a pair only while its wavefront stays clear of the
// band's outermost slots; what it hands on has cost its work for nothing.  At 2 % divergence about half of the 150 bp pairs outgrow the
// 16 slots (and the cascade is faster without the stage), at <= 1 % few do (and it is ~2x faster with it): a pilot on 8192 sampled pairs
// decides once per batch, when the batch is created (b->laneh_pick: 1 = first stage, 2 = not used).
static int pilot_lane_heur(wfa_hip_aligner* al, wfa_hip_batch* b, hipStream_t stream) {
  if (b->laneh_pick != 0 || b->cfg.scope == WFA_SCOPE_FULL) return WFA_HIP_OK;
  const unsigned rtc_failures = wfa::rtc_failure_count();
  b->laneh_pick = 2; b->segh_pick = 2;
  int X, OE, E;
  const bool lane_ok = wfa::lane_heur_config(b->dcfg, b->ncomp);
  const bool seg_ok = wfa::seg_heur_config(b->dcfg, b->ncomp);
  if ((!lane_ok && !seg_ok) || wfa::seg_supported(b->dcfg, b->ncomp, false) ||
      wfa::seg_shape(b->dcfg, &X, &OE, &E) < 0 || b->max_len > WFA_FAST_MAX_LEN || knob(al, K_NO_FAST, 0) != 0) return WFA_HIP_OK;
  const bool free_begins = b->dcfg.endsfree && (b->dcfg.pbf | b->dcfg.tbf) != 0;
  int forced = knob(al, K_LANE_HEUR, -1);       // (WFA_HIP_LANE_HEUR = 1 / 0: always / never, whatever the batch)
  if (!lane_ok) forced = 0;                     // (X-drop: the segmented form only)
  if (wfa::seg_shape(b->dcfg, &X, &OE, &E) == WFA_SHAPE_RTC && !wfa::rtc_lane_shape_ok(X, OE, E)) forced = 0;   // (a run-time shape whose rings outgrow the lane kernel's registers)
  const int forced_seg = knob(al, K_SEG_HEUR, -1);    // (WFA_HIP_SEG_HEUR likewise)
  if (forced >= 0) b->laneh_pick = forced == 2 ? 3 : forced ? 1 : 2;   // (2: the 32-diagonal form)
  if (forced_seg >= 0) b->segh_pick = (forced_seg && seg_ok) ? 1 : 2;
  else if (seg_ok && !free_begins) b->segh_pick = 1;  // (small batches: on, unless wavefront 0 already spans many diagonals)
  if ((forced >= 0 && (forced_seg >= 0 || !seg_ok)) || b->n_packed < 65536u) return WFA_HIP_OK;
  const uint32_t np = 8192u, stride = b->n_packed / np;
  uint32_t* plist = b->d_fb_list2[0];
  uint32_t* psample = b->d_fb_list2[1];
  uint32_t* pcount = b->d_counters + 4;
  hipLaunchKernelGGL(wfa_pilot_sample_kernel, dim3((np + 255u) / 256u), dim3(256), 0, stream, b->d_list_packed, stride, np, psample);
  HIP_TRY(al, hipGetLastError());
  HIP_TRY(al, hipMemsetAsync(pcount, 0, 3 * sizeof(uint32_t), stream));
  wfa::FastArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.words = b->d_words; fa.meta = b->d_meta; fa.worklist = psample; fa.nwork = np;
  fa.score = b->d_score; fa.status = b->d_status; fa.fb_list = plist; fa.fb_count = pcount;
  fa.g = wfa::gcd_int(wfa::gcd_int(b->dcfg.x, b->dcfg.o1 + b->dcfg.e1), b->dcfg.e1);
  fa.ef = (b->dcfg.endsfree && (b->dcfg.pbf | b->dcfg.pef | b->dcfg.tbf | b->dcfg.tef)) ? 1 : 0;
  fa.pbf = b->dcfg.pbf; fa.pef = b->dcfg.pef; fa.tbf = b->dcfg.tbf; fa.tef = b->dcfg.tef;
  fa.heur = b->dcfg.heuristic; fa.min_wf_len = b->dcfg.min_wf_len; fa.max_dist_thr = b->dcfg.max_dist_thr;
  fa.steps_between = b->dcfg.steps_between; fa.max_steps = b->dcfg.max_steps; fa.xdrop = b->dcfg.xdrop; fa.scope = b->dcfg.scope;
  if (forced < 0 &&
      wfa::launch_lane_args(wfa::seg_shape(b->dcfg, &X, &OE, &E), OE, E, al->cu_count, knob(al, K_LANE_WAVES_PER_CU, 48), knob(al, K_LANE_REFILL_MIN, 8),
                            b->max_len, stream, fa, false, 0, 256, 1, X) != 0) return pilot_launch_failed(al, b, rtc_failures);
  // (round 6) ... and through the 32-diagonal form of the same kernel (rings of 16 registers: shallow shapes only)
  const bool lane32_ok = std::max(X, OE) <= 8 && E <= 3 && knob(al, K_LANE_HEUR32, 1) != 0;
  if (forced < 0 && lane32_ok) {
    fa.fb_count = pcount + 2;
    if (wfa::launch_lane_args(wfa::seg_shape(b->dcfg, &X, &OE, &E), OE, E, al->cu_count, knob(al, K_LANE_WAVES_PER_CU, 48), knob(al, K_LANE_REFILL_MIN, 8),
                              b->max_len, stream, fa, false, 0, 256, 2, X) != 0) return pilot_launch_failed(al, b, rtc_failures);
  }
  if (forced_seg < 0 && seg_ok) {   // (the same sample through the 32-lane form; its list is not read, only its count)
    fa.fb_count = pcount + 1;
    if (wfa::launch_seg_heur(b->dcfg, al->cu_count, knob(al, K_FAST_WAVES_PER_CU, 256), stream, fa) != 0) return pilot_launch_failed(al, b, rtc_failures);
  }
  uint32_t handed[3] = {0, 0, 0};
  HIP_TRY(al, hipMemcpyAsync(handed, pcount, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY(al, hipStreamSynchronize(stream));
  HIP_TRY(al, hipMemsetAsync(pcount, 0, 3 * sizeof(uint32_t), stream));
  if (knob(al, K_STAGE_TIMING, 0)) fprintf(stderr, "[wfa_hip] pilot (general forms): 16 slots hand on %u, 32 slots %u, 32-lane segments %u of %u\n", handed[0], handed[2], handed[1], np);
  // (round 6: measured per divergence and configuration, tools/probes/laneh_probe.py — wf-adaptive, free ends of 3 and 12 diagonals, a step
  // limit; 2 M x 150 bp.  16 slots first while they hand on <= 1/4; beyond that 32 slots — 16 registers per component, the price of two
  // 16-slot steps — unless they too hand on > 5 %: then the 32-lane segments start, whose lanes are diagonals and not pairs)
  if (forced < 0) b->laneh_pick = (handed[0] * 4u <= np) ? 1 : (lane32_ok && handed[2] * 20u <= np) ? 3 : 2;
  // (X-drop keeps whole wavefronts: what outgrows the 32 diagonals are the expensive pairs, and the stage pays only below ~1/8 handed on —
  // 150 bp at 2 %, xdrop 100: 21 % handed on, 7.0 ms with the stage, 5.9 ms without)
  if (forced_seg < 0 && seg_ok) b->segh_pick = (handed[1] * (b->dcfg.heuristic == WFA_HEUR_XDROP ? 8u : 4u) <= np) ? 1 : 2;
  return WFA_HIP_OK;
}

static wfa_hip_batch* batch_create_nosync(wfa_hip_aligner_t* al, int64_t n, const uint8_t* seqs,
                                          const int64_t* p_off, const int32_t* p_len,
                                          const int64_t* t_off, const int32_t* t_len, bool in2bit = false) {
  if (!al) { g_error = "null aligner"; return nullptr; }
  if (n < 0 || n > 0x7FFFFFF0ll || (n > 0 && (!seqs || !p_off || !p_len || !t_off || !t_len))) {
    al->err = "invalid batch arguments"; g_error = al->err; return nullptr;
  }
  if (hipSetDevice(al->device) != hipSuccess) { al->err = "hipSetDevice failed"; g_error = al->err; return nullptr; }
  wfa_hip_batch* b = new wfa_hip_batch();
  b->al = al;
  al->live_batches += 1;
  if (in2bit && al->cfg.wildcard >= 0) { al->err = "2-bit reads cannot hold a wildcard letter"; g_error = al->err; batch_free(b); return nullptr; }
  const int rc = batch_build(al, b, n, seqs, p_off, p_len, t_off, t_len, in2bit);
  if (rc != WFA_HIP_OK) { g_error = al->err; batch_free(b); return nullptr; }
  return b;
}

extern "C" wfa_hip_batch_t* wfa_hip_batch_create(wfa_hip_aligner_t* al, int64_t n, const uint8_t* seqs,
                                                 const int64_t* p_off, const int32_t* p_len,
                                                 const int64_t* t_off, const int32_t* t_len) {
  wfa_hip_batch* b = batch_create_nosync(al, n, seqs, p_off, p_len, t_off, t_len);
  // inputs are borrowed for the call only: their upload must be over before it returns
  if (b && b->uploads_pending) { (void)hipStreamSynchronize(al->stream); b->uploads_pending = false; }
  return b;
}

extern "C" wfa_hip_batch_t* wfa_hip_batch_create_packed2bits(wfa_hip_aligner_t* al, int64_t n, const uint8_t* packed,
                                                             const int64_t* p_off, const int32_t* p_len,
                                                             const int64_t* t_off, const int32_t* t_len) {
  wfa_hip_batch* b = batch_create_nosync(al, n, packed, p_off, p_len, t_off, t_len, true);
  if (b && b->uploads_pending) { (void)hipStreamSynchronize(al->stream); b->uploads_pending = false; }
  return b;
}

extern "C" int wfa_hip_align_batch_packed2bits(wfa_hip_aligner_t* al, int64_t n, const uint8_t* packed,
                                               const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len,
                                               int32_t* score, int32_t* status, uint8_t* cigar_ops, const int64_t* cigar_off,
                                               int64_t* cigar_begin, int32_t* cigar_len) {
  if (!al) return WFA_HIP_EINVAL;
  if (al->cfg.wildcard >= 0) { al->err = "2-bit reads cannot hold a wildcard letter"; return WFA_HIP_ENOTSUP; }
  wfa_hip_batch_t* b = batch_create_nosync(al, n, packed, p_off, p_len, t_off, t_len, true);
  if (!b) return (al->err.find("failed:") != std::string::npos) ? WFA_HIP_EDEVICE : WFA_HIP_EINVAL;
  int rc = wfa_hip_batch_run(b, nullptr);
  if (rc == WFA_HIP_OK) rc = wfa_hip_batch_results(b, score, status, cigar_ops, cigar_off, cigar_begin, cigar_len);
  wfa_hip_batch_destroy(b);
  return rc;
}

// hand-over count of a stage as it stands between two of its launches (the pipelined tail of batch_run_once)
__global__ void wfa_snapshot_kernel(const uint32_t* src, uint32_t* dst) { *dst = *src; }

// ---- the walks of a split stage on a stream of their own ----------------------------------------------
// Launch i of a split stage = alignment kernel (history into half i & 1 of the workspace) + the walks of its pairs.  The
// walks are latency-bound chains with little parallelism (one thread per alignment): they run on the side stream, under the
// alignment kernel of the next launch, which goes to the other half.  (Alternating whole launches between two streams
// measured worse: two alignment kernels at once, C4 -35 %.)
struct DualStream {
  wfa_hip_aligner* al; hipStream_t main; int eb = 0;   // eb: this user's pair of events (the band stages' walks and the lane-full stage's expands no longer share them)
  bool on = false; bool used[2] = {false, false};
  int begin(bool want) {
    on = false; used[0] = used[1] = false;
    if (!want || knob(al, K_NO_DUAL, 0) != 0) return WFA_HIP_OK;
    if (!al->band_event[0]) {
      for (int i = 0; i < 4; ++i) {
        HIP_TRY(al, hipEventCreateWithFlags(&al->band_event[i], hipEventDisableTiming));
        HIP_TRY(al, hipEventCreateWithFlags(&al->walk_event[i], hipEventDisableTiming));
      }
    }
    on = true;
    return WFA_HIP_OK;
  }
  // before the alignment kernel of `launch` is enqueued on main: its half must be free (the walks of launch - 2 are over)
  int before_align(int64_t launch) {
    if (on && used[launch & 1]) HIP_TRY(al, hipStreamWaitEvent(main, al->walk_event[eb + (launch & 1)], 0));
    return WFA_HIP_OK;
  }
  // the stream for the walks of `launch` (call after its alignment kernel was enqueued on main)
  int walk_stream(int64_t launch, hipStream_t* out) {
    *out = main;
    if (!on) return WFA_HIP_OK;
    HIP_TRY(al, hipEventRecord(al->band_event[eb + (launch & 1)], main));
    HIP_TRY(al, hipStreamWaitEvent(al->side_stream, al->band_event[eb + (launch & 1)], 0));
    *out = al->side_stream;
    return WFA_HIP_OK;
  }
  int after_walk(int64_t launch) {
    if (!on) return WFA_HIP_OK;
    HIP_TRY(al, hipEventRecord(al->walk_event[eb + (launch & 1)], al->side_stream));
    used[launch & 1] = true;
    return WFA_HIP_OK;
  }
  // join: `main` continues after every walk
  int end() {
    if (!on) return WFA_HIP_OK;
    for (int i = 0; i < 2; ++i) if (used[i]) HIP_TRY(al, hipStreamWaitEvent(main, al->walk_event[eb + i], 0));
    on = false;
    return WFA_HIP_OK;
  }
};

// ---- launch geometry -------------------------------------------------------------------------------

static int ensure_ws(wfa_hip_aligner* al, size_t bytes) {
  if (bytes <= al->ws_bytes) return WFA_HIP_OK;
  if (al->ws) { (void)hipFree(al->ws); al->ws = nullptr; al->ws_bytes = 0; }
  hipError_t e = hipMalloc((void**)&al->ws, bytes);
  if (e == hipErrorOutOfMemory && al->pool_cached > 0) {   // idle blocks of finished batches: give them back and try once more
    (void)hipGetLastError();
    pool_drain_cached(al);
    e = hipMalloc((void**)&al->ws, bytes);
  }
  HIP_TRY(al, e);
  al->ws_bytes = bytes;
  return WFA_HIP_OK;
}

// memory_mode medium / low: the general kernel keeps the piggy-back history (one byte of origin codes per cell) instead of the
// explicit arena (full scope, gap-affine / gap-affine-2p; WFA_HIP_GENERAL_PB = 0 / 1 overrides)
// history of the general kernel: one byte of origin codes per cell (PB) instead of the offsets — memory modes medium / low, and, like
// the banded kernel's split stage, reads over 20 kb in every mode (round 3: the explicit arena of a 100 kb pair overflows and is re-run 8x
// larger, with a host round trip each time: C5-shaped wf-adaptive 10.8 k -> 16 k aln/s); WFA_HIP_BAND_PB=0 / WFA_HIP_GENERAL_PB=0 keep
// the explicit offsets
static bool general_pb(const wfa_hip_aligner* al, const wfa_hip_config_t& c, int ncomp, int max_len) {
  if (c.scope != WFA_SCOPE_FULL || ncomp < 3) return false;
  const int e = knob(al, K_GENERAL_PB, -1);
  if (e >= 0) return e != 0;
  if (c.memory_mode == WFA_MEM_MED || c.memory_mode == WFA_MEM_LOW) return true;
  return max_len > 20000 && knob(al, K_BAND_PB, 1) != 0;   // (at 10 kb the explicit arena fits and its walk is quicker: a lone leftover pair of C4-adaptive 30 ms sooner)
}

static int launch_general_dyn(wfa_hip_aligner* al, wfa_hip_batch* b, hipStream_t stream, bool packed,
                              const uint32_t* worklist, const uint32_t* nwork_dev, uint32_t nwork_host,
                              int64_t ws_stride, int grid, int threads, uint32_t* ovf_list, uint32_t* ovf_count,
                              int32_t* ws_base = nullptr, const uint32_t* wbeg_dev = nullptr) {
  WfaKernelArgs a;
  memset(&a, 0, sizeof(a));
  a.words = b->d_words; a.bytes = b->d_bytes; a.meta = b->d_meta; a.p_boff = b->d_pboff; a.t_boff = b->d_tboff;
  a.worklist = worklist; a.nwork_dev = nwork_dev; a.nwork = nwork_host;
  a.score = b->d_score; a.status = b->d_status;
  a.cigar_ops = b->d_ops; a.cigar_off = b->d_cigar_off; a.cigar_begin = b->d_cigar_begin; a.cigar_len = b->d_cigar_len;
  a.ws = ws_base ? ws_base : al->ws; a.ws_stride = ws_stride;
  a.wbeg_dev = wbeg_dev;
  a.fb_list = ovf_list; a.fb_count = ovf_count;
  a.cfg = b->gcfg;
  if (!packed && b->wild >= 0) a.cfg.wildcard = b->wild;   // (the 8-bit pairs of a batch whose 2-bit pairs run without the wildcard rule)
  if (a.cfg.biwfa_top) a.cfg.heuristic = WFA_HEUR_NONE;   // (standing in for a BiWFA base case: the base aligner has no heuristic, R/wavefront_bialigner.c:66-68)
  if (wfa::launch_general_any(b->gncomp, packed, b->cfg.scope == WFA_SCOPE_FULL, general_pb(al, b->cfg, b->gncomp, b->max_len), a, grid, threads, stream) != 0) {
    al->err = std::string("general kernel launch failed: ") + hipGetErrorString(hipGetLastError());
    return WFA_HIP_EDEVICE;
  }
  return WFA_HIP_OK;
}

// Geometry of the general kernel: threads per alignment, workgroups in flight, workspace per workgroup.
struct Geometry { int threads; int grid; int64_t ws_stride; };

static int64_t free_budget(wfa_hip_aligner* al) {
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = al->total_mem / 2;
  // what we already hold as workspace can be re-used, and so can the idle blocks of the pool (drained on demand)
  return (int64_t)((double)(fr + al->ws_bytes + al->pool_cached) * 0.8);
}

// The history of a full-CIGAR run is sized for scores up to 0.9 x the read length under pywfa's default penalties (4/6/2: about 15 %
// divergence); other penalties (also the rescaled ones of match < 0) scale the score a divergence costs
static double penalty_scale(const WfaDevConfig& d) {
  return std::max(1.0, std::max(d.x / 4.0, std::max((d.o1 + d.e1) / 8.0, d.e1 / 2.0)));
}

static Geometry plan_general(wfa_hip_aligner* al, const wfa_hip_batch* b, uint32_t nwork, int64_t arena_ints) {
  Geometry g;
  const bool full = (b->cfg.scope == WFA_SCOPE_FULL);
  int threads = 64;
  if (b->max_len > 2000) threads = (b->cfg.heuristic == WFA_HEUR_ADAPTIVE) ? 64 : 256;
  threads = knob(al, K_THREADS, threads);
  threads = std::max(64, std::min(512, (threads / 64) * 64));
  const int waves = threads / 64;
  int per_cu = std::max(1, knob(al, K_WAVES_PER_CU, 32) / waves);
  int64_t grid = (int64_t)al->cu_count * per_cu;
  grid = std::min<int64_t>(grid, std::max<uint32_t>(nwork, 1));
  int64_t stride;
  if (full) stride = arena_ints;
  else stride = (int64_t)b->dcfg.scope * b->ncomp * b->max_width;
  stride = (stride + 63) & ~63ll;
  const int64_t budget = free_budget(al);
  while (grid > 1 && grid * stride * 4 > budget) grid = (grid + 1) / 2;
  g.threads = threads; g.grid = (int)grid; g.ws_stride = stride;
  return g;
}

static int64_t initial_arena_ints(const wfa_hip_aligner* al, const wfa_hip_batch* b) {
  typedef long long ll;
  const int mi = 2 * b->ncomp + 4;
  // room for a run whose score is ~ the longer sequence and whose wavefronts are ~128 wide, and never
  // less than what wavefront 0 and a few hundred scores need; overflowing pairs are re-run larger
  ll ints = (ll)b->max_len * (b->ncomp * 64 + mi) / 4 + (ll)b->max_width * b->ncomp * 4 + 4096 * mi;
  // piggy-back history: one byte per cell + 12 bytes per score (the ring of offsets is b->arena_fixed)
  if (general_pb(al, b->cfg, b->gncomp, b->max_len)) ints = (ll)b->max_len * (64 + 12) / 4 + 4096;
  ints = std::max<ll>(ints, 1 << 14);
  const int e = knob(al, K_ARENA_KB, 0);
  if (e > 0) ints = (ll)e * 256;
  return ints;
}

static int batch_run_once(wfa_hip_batch_t* b, void* stream_);

// A run whose penalties have no instantiated kernel launches kernels compiled at run time.  If one of them cannot be built, loaded or
// launched (a hipRTC / comgr failure, a shape the compiler rejects), the run is planned again without the run-time path — the tiled
// and general kernels take those penalties, slower, with the same results — and the compiler's message is kept in the aligner's
// error text (such a shape used to run on the general kernel before the run-time path existed, and must still run).
extern "C" int wfa_hip_batch_run(wfa_hip_batch_t* b, void* stream_) {
  if (!b) return WFA_HIP_EINVAL;
  const unsigned failures = wfa::rtc_failure_count();
  const size_t ev_used = b->ev_used;
  const int runs_pending = b->runs_pending;
  int rc = batch_run_once(b, stream_);
  if (rc == WFA_HIP_EDEVICE && b->dcfg.rtc && wfa::rtc_failure_count() != failures) {
    wfa_hip_aligner* al = b->al;
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream_ ? (hipStream_t)stream_ : al->stream);   // (what the failed run had enqueued before it gave up)
    if (al->side_stream) (void)hipStreamSynchronize(al->side_stream);
    if (al->tail_stream) (void)hipStreamSynchronize(al->tail_stream);
    b->ev_used = ev_used; b->runs_pending = runs_pending;   // (the failed run's timing events: its end was never recorded)
    b->dcfg.rtc = 0; al->dcfg.rtc = 0;
    al->rtc_note = std::string("run-time kernels switched off for this aligner: ") + wfa::rtc_last_error();
    rc = batch_run_once(b, stream_);
    if (rc != WFA_HIP_OK) al->err += " (" + al->rtc_note + ")";
  }
  return rc;
}

static int batch_run_once(wfa_hip_batch_t* b, void* stream_) {
  wfa_hip_aligner* al = b->al;
  HIP_TRY(al, hipSetDevice(al->device));
  if (al->mb_h && __atomic_load_n(&al->mb_h->alive, __ATOMIC_ACQUIRE) != 0) mailbox_quit(al);
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : al->stream;
  // every run of this aligner uses the one workspace (al->ws): order this run after the previous one when it was
  // enqueued on another stream
  if (al->ws_event_recorded && al->ws_last_stream != stream) HIP_TRY(al, hipStreamWaitEvent(stream, al->ws_event, 0));
  // the batch's uploads were enqueued on the aligner's stream (and may still be in flight): a run elsewhere waits for them
  if (stream != al->stream && b->upload_event) HIP_TRY(al, hipStreamWaitEvent(stream, b->upload_event, 0));
  // an early return (an error) must not leave walks / expands running on the side stream: they write blocks the caller may
  // destroy next
  struct SideGuard {
    wfa_hip_aligner* al; bool ok = false;
    ~SideGuard() {
      if (!ok && al->side_stream) (void)hipStreamSynchronize(al->side_stream);
      if (!ok && al->tail_stream) (void)hipStreamSynchronize(al->tail_stream);
    }
  } side_guard{al};
  b->last_stream = stream;
  b->ran = true; b->synced = false;
  b->last_fallback = 0;
  b->last_kernel_pairs = 0;
  if (b->n == 0) { b->synced = true; side_guard.ok = true; return WFA_HIP_OK; }
  const bool full = (b->cfg.scope == WFA_SCOPE_FULL);
  HIP_TRY(al, hipMemsetAsync(b->d_counters, 0, WFA_COUNTER_WORDS * sizeof(uint32_t), stream));   // (every stage's hand-over count too: one fill per run, not one per stage)
  b->arena_ints = full ? initial_arena_ints(al, b) : 0;
  b->arena_fixed = general_pb(al, b->cfg, b->gncomp, b->max_len) ? (((int64_t)b->dcfg.scope * b->ncomp * b->max_width + 64 + 63) & ~63ll) : 0;

  // A cascade of kernels over the 2-bit pairs: each stage aligns what fits it and appends the rest to
  // a leftover list (pair ids + a device-side count) that the next stage consumes on the same stream;
  // the general kernel at the end takes everything.  All stages compute the same wavefronts, so which
  // stage finishes a pair does not change its result.
  while (b->ev.size() < b->ev_used + 2) { hipEvent_t e; HIP_TRY(al, hipEventCreate(&e)); b->ev.push_back(e); }
  hipEvent_t ev0 = b->ev[b->ev_used], ev1 = b->ev[b->ev_used + 1];
  b->ev_used += 2; b->runs_pending += 1;
  HIP_TRY(al, hipEventRecord(ev0, stream));
  const bool biwfa_score = !full && b->cfg.memory_mode == WFA_MEM_BIWFA && (b->cfg.max_steps > 0 || b->cfg.heuristic != WFA_HEUR_NONE);
  if ((full && b->cfg.memory_mode == WFA_MEM_BIWFA) || biwfa_score) {
    // BiWFA: one wave per alignment; a workgroup's slice of the workspace = forward ring + reverse ring + base-case history.
    // (scope=score with a step limit: the top-level breakpoint search alone, no base-case history)
    wfa::BiwfaArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.score_only = biwfa_score ? 1 : 0;
    ba.ring_stride = b->max_width;
    ba.ring_ints = ((int64_t)b->dcfg.scope * b->ncomp * ba.ring_stride + 63) & ~63ll;
    ba.base_stride = wfa::biwfa_base_stride(b->max_width);
    ba.base_ints = biwfa_score ? 0 : ((wfa::biwfa_base_ints(b->ncomp, ba.base_stride) + 63) & ~63ll);
    const int64_t stride = 2 * ba.ring_ints + ba.base_ints;
    int64_t grid = std::min<int64_t>((int64_t)al->cu_count * knob(al, K_WAVES_PER_CU, 32), std::max<int64_t>(b->n, 1));
    const int64_t budget = free_budget(al);
    while (grid > 1 && grid * stride * 4 > budget) grid = (grid + 1) / 2;
    // the general kernel behind it (full scope): pairs whose top-level base case outgrew the BiWFA kernel's history
    // (reads of <= 100 bases under large penalties) are aligned by the ordinary algorithm, which is what that base case is
    b->dcfg.biwfa_top = full ? 1 : 0;
    b->gcfg.biwfa_top = b->dcfg.biwfa_top;   // (the general kernel's copy of the configuration)
    Geometry g = plan_general(al, b, (uint32_t)std::min<int64_t>(b->n, (int64_t)al->cu_count * 16), b->arena_fixed + b->arena_ints);
    // Round 5: full CIGARs go level by level (csrc/wfa_bilevel.hpp: every window of a recursion level is a work item of one launch);
    // the depth-first kernel keeps the score-only form and what the level queues could not hold (redo list).
    //   workspace: [ring / base-history slices][window queues, leaves, per-pair words, counters][slices of the depth-first kernel]
    wfa::BlArgs la;
    memset(&la, 0, sizeof(la));
    const bool bilevel = full && knob(al, K_BILEVEL, 1) != 0;
    const bool bl_i16 = b->max_len <= 32000 && knob(al, K_BILEVEL_I32, 0) == 0;
    int bl_grid = 0, bl_levels = 0;
    int64_t dfs_off = 0;
    size_t bl_bytes = 0;
    if (bilevel) {
      const int64_t esz = bl_i16 ? 2 : 4;
      la.ring_stride = (b->max_width + 1) & ~1;
      la.ring_elems = ((int64_t)b->dcfg.scope * b->ncomp * la.ring_stride + 63) & ~63ll;
      la.base_stride = ba.base_stride; la.base_ints = ba.base_ints;
      la.slice_bytes = (std::max<int64_t>(2 * la.ring_elems * esz, ba.base_ints * 4) + 255) & ~255ll;
      bl_grid = al->cu_count * std::max(1, knob(al, K_BILEVEL_PER_CU, 32));
      // (a slice is megabytes: no more of them than the batch can keep busy — a few windows per pair are in flight at the deep levels)
      bl_grid = (int)std::min<int64_t>(bl_grid, std::max<int64_t>(64, 8 * (int64_t)std::max<int64_t>(b->n_packed, b->n_bytes)));
      // windows a pair can be in at once: about two per 250 of score; queues sized from the longest pair, what overflows is redone
      const int64_t per_pair = std::min<int64_t>(1024, std::max<int64_t>(4, b->max_width / 96));
      const int64_t nmax = std::max<int64_t>(b->n_packed, b->n_bytes);
      int64_t qcap = knob(al, K_BILEVEL_QCAP, 0) > 0 ? knob(al, K_BILEVEL_QCAP, 0) : std::min<int64_t>(nmax * per_pair + 1024, (int64_t)1 << 28);   // (the knob: tests of the redo path)
      const int64_t pair_bytes = (int64_t)b->n * (4 + 4 + 4 + 8 + 4) + WFA_BL_COUNTER_WORDS * 4 + 4096;
      // the depth-first kernel behind it: a few slices (its int32 rings are the large ones)
      grid = std::min<int64_t>(std::min<int64_t>(grid, al->cu_count), std::max<int64_t>(1, nmax));   // (the redo list holds pairs of this batch)
      // the queues must leave room for one ring slice per CU and one slice of the depth-first kernel — a large batch of
      // short reads (10 M pairs: 6.4 GB of queues) would otherwise ask for more than the device has.  Queue overflow is a speed matter
      // only (the redo list), so the capacity is clamped to half of what remains
      if (knob(al, K_BILEVEL_QCAP, 0) <= 0) {
        const int64_t room = budget - (int64_t)al->cu_count * la.slice_bytes - pair_bytes - stride * 4;
        qcap = std::max<int64_t>(1024, std::min<int64_t>(qcap, room / 2 / 160));
      }
      const int64_t meta_bytes = qcap * 32 * 5 + pair_bytes;
      while (bl_grid > al->cu_count && (int64_t)bl_grid * la.slice_bytes + meta_bytes + grid * stride * 4 > budget) bl_grid /= 2;
      while (grid > 1 && (int64_t)bl_grid * la.slice_bytes + meta_bytes + grid * stride * 4 > budget) grid = (grid + 1) / 2;
      la.qcap = la.qbcap = la.leafcap = la.qwcap = (uint32_t)qcap;
      bl_bytes = (size_t)((int64_t)bl_grid * la.slice_bytes + meta_bytes);
      dfs_off = (int64_t)((bl_bytes + 255) & ~(size_t)255);
      // levels: a window's score halves per level (to within the ring's scope) until it is <= 250
      const int64_t pmax = std::max<int64_t>(1, std::max<int64_t>(b->dcfg.x, std::max<int64_t>(b->dcfg.o1 + b->dcfg.e1, b->dcfg.o2 + b->dcfg.e2)));
      int64_t smax = pmax * (int64_t)b->max_width;
      bl_levels = 3;
      while (smax > 250 && bl_levels < WFA_BL_MAX_LEVELS - 1) { smax /= 2; ++bl_levels; }
      if (knob(al, K_BILEVEL_LEVELS, 0) > 0) bl_levels = std::min(knob(al, K_BILEVEL_LEVELS, 0), WFA_BL_MAX_LEVELS - 1);
    }
    int rc = ensure_ws(al, std::max((size_t)dfs_off + (size_t)grid * stride * 4, full ? (size_t)g.grid * g.ws_stride * 4 : (size_t)0));
    if (rc != WFA_HIP_OK) return rc;
    WfaKernelArgs& a = ba.k;
    a.words = b->d_words; a.bytes = b->d_bytes; a.meta = b->d_meta; a.p_boff = b->d_pboff; a.t_boff = b->d_tboff;
    a.score = b->d_score; a.status = b->d_status;
    a.cigar_ops = b->d_ops; a.cigar_off = b->d_cigar_off; a.cigar_begin = b->d_cigar_begin; a.cigar_len = b->d_cigar_len;
    a.ws = reinterpret_cast<int*>(reinterpret_cast<char*>(al->ws) + dfs_off); a.ws_stride = stride; a.cfg = b->dcfg;
    if (bilevel) {
      char* p = reinterpret_cast<char*>(al->ws);
      la.rings = p; p += (int64_t)bl_grid * la.slice_bytes;
      la.q[0] = reinterpret_cast<wfa::BlWindow*>(p); p += (int64_t)la.qcap * 32;
      la.q[1] = reinterpret_cast<wfa::BlWindow*>(p); p += (int64_t)la.qcap * 32;
      la.qb = reinterpret_cast<wfa::BlWindow*>(p); p += (int64_t)la.qcap * 32;
      la.qw = reinterpret_cast<wfa::BlWindow*>(p); p += (int64_t)la.qcap * 32;
      la.leaves = reinterpret_cast<wfa::BlLeaf*>(p); p += (int64_t)la.qcap * 32;
      la.failkey = reinterpret_cast<unsigned long long*>(p); p += (int64_t)b->n * 8;
      la.head = reinterpret_cast<int*>(p); p += (int64_t)b->n * 4;
      la.flags = reinterpret_cast<int*>(p); p += (int64_t)b->n * 4;
      la.top = reinterpret_cast<int*>(p); p += (int64_t)b->n * 4;
      la.redo_list = reinterpret_cast<uint32_t*>(p); p += (int64_t)b->n * 4;
      la.cnt = reinterpret_cast<uint32_t*>(p);
    }
    for (int kind = 0; kind < 2; ++kind) {   // 2-bit pairs, then the pairs aligned on their bytes
      const uint32_t cnt = kind ? b->n_bytes : b->n_packed;
      if (cnt == 0) continue;
      a.worklist = kind ? b->d_list_bytes : b->d_list_packed; a.nwork_dev = nullptr; a.nwork = cnt;
      a.fb_list = full ? b->d_fb_list2[kind] : nullptr; a.fb_count = b->d_counters + 4 + kind;
      a.cfg.wildcard = (kind && b->wild >= 0) ? b->wild : b->dcfg.wildcard;   // (the 8-bit pairs keep the wildcard rule)
      int64_t dfs_grid = std::min<int64_t>(grid, cnt);
      if (bilevel) {
        la.k = a;
        HIP_TRY(al, hipMemsetAsync(la.cnt, 0, WFA_BL_COUNTER_WORDS * sizeof(uint32_t), stream));
        const bool stage_timing = knob(al, K_STAGE_TIMING, 0) != 0;   // development aid: synchronises
        std::vector<hipEvent_t> tev;
        auto mark = [&]() { if (stage_timing) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, stream); tev.push_back(e); } };
        mark();
        bool ok = wfa::launch_bl_seed(la, stream) == 0;
        // four waves per window while the wavefronts are expected to span several chunks (about max_len / 8 diagonals at the top
        // level for reads at 8 - 10 %, half of it per level)
        int wide_auto = 0;
        while (wide_auto < 8 && ((b->max_len / 8) >> wide_auto) >= 512) ++wide_auto;   // (10 kb: two levels, measured best of 0 / 2 / 3)
        const int wide_levels = knob(al, K_BILEVEL_WIDE_LEVELS, wide_auto);
        // Per level: the LDS form first where it pays (2-bit pairs, int16 offsets, gap-affine and the one-component metrics): both
        // aligners' rows and the window's sub-sequences in LDS, W diagonals per row — W halves per level as the windows' scores do;
        // used where at least two windows fit a CU (W <= 1024 for gap-affine 4/6/2).  A window whose wavefront outgrows W (or whose
        // sub-sequences outgrow the buffers) moves to the level's second launch, the workspace form: rows in HBM / L2, the pair's
        // sequences in LDS when they fit, four chunks of loads in flight per thread.
        const bool lds_form = kind == 0 && bl_i16 && b->ncomp <= 3 && knob(al, K_BILEVEL_LDS, 0) != 0;   // (off: measured slower than the workspace form at every level of 1 kb and 10 kb reads — per-step overheads, not the rows' latency, are what a narrow window pays)
        const int gsc = std::max(1, wfa::band_gcd(b->dcfg, false));
        la.lds_slots = (b->dcfg.scope - 1) / gsc + 1;
        const int full_seq_words = (b->max_len + 15) / 16 + 4;
        const bool seql = kind == 0 && (size_t)2 * full_seq_words * 4 <= (size_t)48 * 1024 && knob(al, K_BILEVEL_NO_SEQL, 0) == 0;
        const int lds_w_max = knob(al, K_BILEVEL_LDS_W, 1024);
        for (int lv = 0; lv < bl_levels && ok; ++lv) {
          la.level = lv;
          la.from_wide = 0;
          // expected width of a level's wavefronts: a quarter of the window's length at 8 - 10 % error; rows of 1.3 x that, a power of two
          int w = 128;
          while (w < 8192 && w < (int)((((int64_t)b->max_len * 13 / 40) >> lv))) w *= 2;
          if (lds_form && w <= lds_w_max) {
            la.lds_seq_words = std::min(full_seq_words, (int)(((int64_t)b->max_len * 3 / 2) >> lv) / 16 + 12);
            const int lthreads = (w >= 2048) ? 1024 : (w >= 512) ? 256 : 64;
            const size_t lsm = wfa::bl_split_lds_smem(b->ncomp, b->dcfg.scope, lthreads, w, la.lds_slots, la.lds_seq_words);
            if (lsm <= (size_t)156 * 1024) {
              la.lds_w = w;
              const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / (lsm + 512), (size_t)(2048 / lthreads)));
              const int64_t lgrid = (int64_t)al->cu_count * per_cu;
              ok = wfa::launch_bl_split_lds_any(b->ncomp, lthreads, la, (int)std::min<int64_t>(lgrid, lv == 0 ? (int64_t)cnt : lgrid), stream) == 0;
              la.from_wide = 1;
            }
          }
          // (expected width of a level's wavefronts: a quarter of the read length at the top, half of it per level; sixteen waves per
          // window from 8 192 diagonals on (measured at 20 / 40 / 100 kb: 1 024 / 2 048 / 4 096 / 8 192): 100 kb reads' top levels — a step is then 6 passes over the wavefront instead of 24)
          const int64_t w_est = ((int64_t)b->max_len / 4) >> lv;
          const bool few_windows = ((int64_t)cnt << std::min(lv, 20)) <= 2 * (int64_t)al->cu_count;   // (every window of the level resident at sixteen waves each)
          const bool huge = kind == 0 && knob(al, K_BILEVEL_NO_1024, 0) == 0 &&
                            (w_est >= knob(al, K_BILEVEL_HUGE_MIN, 8192) || (few_windows && w_est >= 2048));   // (small batches: 64 / 256 x 10 kb +20 %)
          const int threads = huge ? 1024 : (lv < wide_levels) ? 256 : 64;
          const int lgrid = huge ? std::min(bl_grid, 2 * al->cu_count)
                                 : (threads == 256) ? std::min(bl_grid, std::max(al->cu_count, bl_grid / 2)) : bl_grid;   // (a slice of the workspace per workgroup: never more than bl_grid)
          la.lds_seq_words = full_seq_words;
          ok = ok && wfa::launch_bl_split_any(b->ncomp, kind == 0, bl_i16, threads, seql, la, (int)std::min<int64_t>(lgrid, lv == 0 ? (int64_t)cnt : (int64_t)lgrid), stream) == 0;
          la.from_wide = 0;
          mark();
        }
        ok = ok && wfa::launch_bl_base_any(b->ncomp, kind == 0, la, bl_grid, stream) == 0;
        mark();
        ok = ok && wfa::launch_bl_finish(la, (int)std::min<int64_t>((int64_t)al->cu_count * 16, cnt), stream) == 0;
        mark();
        if (!ok) { al->err = "BiWFA level kernel launch failed"; return WFA_HIP_EDEVICE; }
        a.worklist = la.redo_list; a.nwork_dev = la.cnt + WFA_BL_MAX_LEVELS + 2; a.nwork = cnt;
        dfs_grid = grid;
        if (wfa::launch_biwfa_any(b->ncomp, kind == 0, ba, (int)dfs_grid, stream) != 0) { al->err = "BiWFA kernel launch failed"; return WFA_HIP_EDEVICE; }
        mark();
        if (stage_timing) {
          (void)hipStreamSynchronize(stream);
          uint32_t hc[WFA_BL_COUNTER_WORDS];
          (void)hipMemcpy(hc, la.cnt, sizeof(hc), hipMemcpyDeviceToHost);
          fprintf(stderr, "[wfa_hip] biwfa levels (%s, %d pairs, grid %d, slice %.2f MB, %s rings):", kind ? "bytes" : "2-bit", (int)cnt, bl_grid, la.slice_bytes / 1048576.0, bl_i16 ? "int16" : "int32");
          for (int lv = 0; lv < bl_levels; ++lv) { float ms = 0; (void)hipEventElapsedTime(&ms, tev[lv], tev[lv + 1]); fprintf(stderr, " L%d %u windows (%u past the LDS form) %.3f ms;", lv, hc[lv], hc[128 + lv], ms); }
          float mb = 0, mf = 0, md = 0;
          (void)hipEventElapsedTime(&mb, tev[bl_levels], tev[bl_levels + 1]); (void)hipEventElapsedTime(&mf, tev[bl_levels + 1], tev[bl_levels + 2]); (void)hipEventElapsedTime(&md, tev[bl_levels + 2], tev[bl_levels + 3]);
          fprintf(stderr, " base %u windows %.3f ms; finish %.3f ms (%u leaves); redo %u pairs %.3f ms\n", hc[WFA_BL_MAX_LEVELS], mb, mf, hc[WFA_BL_MAX_LEVELS + 1], hc[WFA_BL_MAX_LEVELS + 2], md);
          for (hipEvent_t e : tev) (void)hipEventDestroy(e);
        }
        continue;
      }
      if (wfa::launch_biwfa_any(b->ncomp, kind == 0, ba, (int)dfs_grid, stream) != 0) { al->err = "BiWFA kernel launch failed"; return WFA_HIP_EDEVICE; }
    }
    if (full) {
      for (int kind = 0; kind < 2; ++kind) {
        if ((kind ? b->n_bytes : b->n_packed) == 0) continue;
        rc = launch_general_dyn(al, b, stream, kind == 0, b->d_fb_list2[kind], b->d_counters + 4 + kind, 0u, g.ws_stride, g.grid, g.threads,
                                b->d_ovf_list[0], b->d_counters + 1);
        if (rc != WFA_HIP_OK) return rc;
      }
    }
    b->last_kernel_pairs = b->n;
    HIP_TRY(al, hipEventRecord(ev1, stream));
    HIP_TRY(al, hipEventRecord(al->ws_event, stream));
    al->ws_event_recorded = true; al->ws_last_stream = stream;
    side_guard.ok = true;
    return WFA_HIP_OK;
  }
  if (b->n_packed > 0) {
    const uint32_t* in_list = b->d_list_packed;   // nullptr = identity
    const uint32_t* in_count = nullptr;            // nullptr = host count
    uint32_t in_n = b->n_packed;
    int out_sel = 0;                               // leftovers go to d_fb_list[out_sel]; their count: a word of its own per stage
    int stage_counts = 0;                          // (zeroed with all the counters at the start of the run: round 4 — a fill per stage was 9 x 4.5 us of a 3.2 ms C2 step)
    auto next_count = [&]() -> uint32_t* { uint32_t* c = b->d_counters + 16 + stage_counts; if (stage_counts < WFA_COUNTER_WORDS - 17) ++stage_counts; return c; };
    bool first_stage = true;
    const bool adapt = (b->dcfg.heuristic != WFA_HEUR_NONE);  // the heuristic instantiations of the banded kernel (wf-adaptive / X-drop)
    // the general kernel's geometry is fixed first so that one workspace allocation serves every stage
    int n_stages = 0;
    int band_nch[3] = {0, 0, 0};
    // a handful of short pairs: one launch of the general kernel beats six nearly empty stages (single-pair calls of a
    // pywfa-style loop).  Long reads take the stages whatever the count: an alignment of milliseconds dwarfs the launches, and
    // the banded / wide kernels are several times faster per pair (8 pairs of C4 as written: 57 ms instead of 436 ms; 8 x 10 kb
    // wf-adaptive with full CIGAR: 3.9 instead of 15.7 ms)
    const bool tiny = in_n <= (uint32_t)knob(al, K_TINY_BATCH, 128) && b->max_len <= 1000;
    const bool use_fast = !tiny && !full && wfa::seg_supported(b->dcfg, b->ncomp, full) && b->max_len <= WFA_FAST_MAX_LEN &&
                          knob(al, K_NO_FAST, 0) == 0;
    // full CIGARs of short reads: the segmented kernel with a history slot per pair, then the thread-per-alignment walk
    // (round 6: a one-component distance with CIGARs, mapped — b->dcfg.lin: the register stages' LIN form exists as a run-time instantiation
    // only; without hipRTC, and for what those stages hand on, the general kernel under the original configuration)
    const bool lin = b->dcfg.lin != 0;
    const bool lin_regs = lin && b->dcfg.rtc != 0 && wfa::rtc_available();
    const bool use_segfull = (!lin || lin_regs) && !tiny && full && wfa::seg_supported(b->dcfg, b->ncomp, false) && b->max_len <= WFA_FAST_MAX_LEN &&
                             knob(al, K_NO_FAST, 0) == 0 && knob(al, K_NO_SEGFULL, 0) == 0;
    // Round 3: score-only short reads with wf-adaptive, free ends or a step limit — what the bound of the register kernels cannot
    // prove — start in the general form of the lane kernel (wfa_lane_kernel<.., HEUR>: the pair is handed on the moment its
    // wavefront touches the band's outermost slots); the banded stages take what it hands on
    int lh_x, lh_oe, lh_e;
    const bool use_laneh = !tiny && !full && !use_fast && wfa::lane_heur_config(b->dcfg, b->ncomp) && wfa::seg_shape(b->dcfg, &lh_x, &lh_oe, &lh_e) >= 0 &&
                           (wfa::seg_shape(b->dcfg, &lh_x, &lh_oe, &lh_e) != WFA_SHAPE_RTC || wfa::rtc_lane_shape_ok(lh_x, lh_oe, lh_e)) &&
                           b->max_len <= WFA_FAST_MAX_LEN && knob(al, K_NO_FAST, 0) == 0 &&
                           (b->laneh_pick == 1 || b->laneh_pick == 3 || (b->laneh_pick == 0 && knob(al, K_LANE_HEUR, 0) != 0));   // (the pilot of batch_build, or WFA_HIP_LANE_HEUR=1 / 2)
    const int laneh_form = (b->laneh_pick == 3 || (b->laneh_pick == 0 && knob(al, K_LANE_HEUR, 0) == 2)) ? 2 : 1;   // (2: 32 diagonals per pair)
    // ... and then (or first, when the lane form's pilot said no) the same form of the 32-lane segments: two pairs per wave, a band twice
    // as wide (WFA_HIP_SEG_HEUR=0: off)
    const bool use_segh = !tiny && !full && !use_fast && wfa::seg_heur_config(b->dcfg, b->ncomp) && b->max_len <= WFA_FAST_MAX_LEN &&
                          knob(al, K_NO_FAST, 0) == 0 && !wfa::seg_supported(b->dcfg, b->ncomp, false) &&
                          (b->segh_pick == 1 || (b->segh_pick == 0 && knob(al, K_SEG_HEUR, 0) != 0)) &&   // (the pilot of batch_build, or WFA_HIP_SEG_HEUR=1)
                          // (behind the 32-slot lane form the segments would see the same 32 diagonals again: what it hands on — 2 % of 150 bp
                          // pairs at 2 % — goes straight to the banded stages, 2.22 -> 2.02 ms per 2 M pairs; WFA_HIP_SEG_HEUR=1 keeps the stage)
                          !(use_laneh && laneh_form == 2 && knob(al, K_SEG_HEUR, -1) < 0);
    if (!lin && !tiny && wfa::band_supported(b->dcfg, b->ncomp) && (b->ncomp != 5 || b->max_len < 32000) && knob(al, K_NO_BAND, 0) == 0) {
      if (b->ncomp == 5) {
        // gap-affine-2p wavefronts are wide (C4: 99 % need more than 108 diagonals, 2.6 % more than 172)
        if (b->band_pick != 2) { band_nch[n_stages++] = 3; band_nch[n_stages++] = 4; }   // (2: the pilot of batch_build saw most pairs outgrow 256 diagonals)
      } else if (adapt) {
        // (the wavefront a cut-off keeps grows with the read: at 10 kb 3 % of the pairs outgrow the 128-diagonal window, at 100 kb 46 % —
        // and a second stage of few, lone waves runs at its pairs' latency: reads over 20 kb start in the 256-diagonal window.
        // 8 192 x 100 kb wf-adaptive: 509 -> 359 ms)
        if (b->max_len <= 300) { band_nch[n_stages++] = 1; }
        if (b->max_len <= 20000) band_nch[n_stages++] = 2;
        band_nch[n_stages++] = 4;
      } else if (b->max_len <= 300) {
        if (!use_fast && !use_segfull) band_nch[n_stages++] = 1;
        band_nch[n_stages++] = 2; band_nch[n_stages++] = 4;
      } else if (b->max_len <= 1200) {
        if (b->band_pick != 2) band_nch[n_stages++] = 4;   // (2: the pilot of batch_build saw most pairs outgrow the window)
      }
      const int only = knob(al, K_BAND_NCH, 0);
      if (only) { n_stages = 1; band_nch[0] = only; }
    }
    // (round 3) wf-adaptive: the wide kernel takes what the banded stages hand on — the few pairs whose wavefront outgrows 256 diagonals
    // (47 of 8 192 at 100 kb) — instead of the general kernel: rows in the workspace, its cut-off in-kernel (WFA_HIP_WIDE_ADAPT=0: off)
    // (reads over 20 kb: at 10 kb the general kernel is as quick for the odd pair, and the 2p rows of the wide kernel are slower)
    const int wide_adapt_knob = knob(al, K_WIDE_ADAPT, 1);
    const bool wide_adapt = b->dcfg.heuristic == WFA_HEUR_ADAPTIVE && b->max_len > 1000 &&
                            (wide_adapt_knob == 2 || (n_stages > 0 && wide_adapt_knob != 0 && b->max_len > 20000));   // (2: any length, also without banded stages in front: tests)
    const bool wide_ok = !lin && !tiny && ((b->ncomp == 3 && b->dcfg.metric == 3) || (b->ncomp == 5 && b->dcfg.metric == 4 && b->dcfg.e2 >= 1)) &&
                         (b->dcfg.heuristic == WFA_HEUR_NONE || wide_adapt) && b->dcfg.match == 0 &&
                         b->max_len > 64 && 2 * (int64_t)b->max_len <= 0x3fffff00ll && b->dcfg.e1 >= 1 && b->dcfg.x >= 1 && knob(al, K_NO_WIDE, 0) == 0;
    // reads beyond 16 kb (plen + tlen > 32 000 does not fit int16 offsets): the workspace-row form with int32 rows (round 3)
    const bool wide32 = 2 * (int64_t)b->max_len > 32000;
    const bool any_pre = use_fast || use_laneh || use_segh || use_segfull || n_stages > 0 || wide_ok;
    Geometry g = plan_general(al, b, any_pre ? std::min<uint32_t>(in_n, (uint32_t)al->cu_count * 16) : in_n, b->arena_fixed + b->arena_ints);
    size_t need = (size_t)g.grid * g.ws_stride * 4;
    // band history: fixed-stride records per score step, one slice per wave
    long long band_grid[3] = {0, 0, 0};
    int64_t band_stride[3] = {0, 0, 0};
    size_t split_region = 0, later_need = 0;   // bytes of the split stage's slots; largest history of the band stages behind it
    // The split stage (long reads) keeps the piggy-back history (the reference's R/wavefront_backtrace_offload.c scheme): one
    // byte of origin codes per (step, diagonal) instead of the offsets, the matches re-extended afterwards.  In EVERY
    // memory mode: the op strings are the same and it is the faster form (C3 91 vs 97 ms per 100 k pairs, C4 adaptive 57 vs
    // 77 ms per 10 k: 8-16x less history, so more pairs per launch); WFA_HIP_BAND_PB=0 keeps the explicit offsets.
    const int pb_env = knob(al, K_BAND_PB, -1);
    const bool pb_mode = full && (pb_env >= 0 ? pb_env != 0 : true);
    int64_t pb_code_ints = 0, pb_event_ints = 0, pb_stride = 0;
    // Round 6, the pipelined tail (16 % of a C4-adaptive run was a latency tail — the 256-diagonal stage 22.7 ms
    // and the general kernel 17.4 ms for ONE pair, both after the last launch of the split stage).  What launch j of the split stage
    // hands on is aligned by the stage behind it — and what THAT hands on by the general kernel — on a third stream beside launch
    // j + 1: [snapshot of the hand-over count before, after) is the launch's part of the list (BandArgs::wbeg_dev).  Only the last
    // launch's leftovers remain exposed.  Needs a region of its own for each of the two (behind the split stage's slots), two band
    // stages, long reads with CIGARs (the split form), no tile / wide stage in the chain.  WFA_HIP_PIPE_TAIL=0: off
    const bool want_pipe_tail = full && n_stages == 2 && b->max_len > 1000 && knob(al, K_PIPE_TAIL, 1) != 0 && knob(al, K_NO_DUAL, 0) == 0 &&
                                knob(al, K_STAGE_TIMING, 0) == 0 && !al->knobs.set[K_BAND_LEFTOVER_WAVES_PER_CU];
    for (int i = 0; i < n_stages; ++i) {
      // 4x more waves than a CU holds at once: waves retire one after the other (oldest-first issue) and the
      // dispatcher refills the CU, instead of a tail of lone waves (C1 +20 %, C3 +13 %)
      long long grid = (long long)al->cu_count * knob(al, K_BAND_WAVES_PER_CU, 128);
      grid = std::min<long long>(grid, in_n);
      if (i > 0 || use_fast || use_laneh || use_segh || use_segfull) grid = std::min<long long>(grid, (long long)al->cu_count * knob(al, K_BAND_LEFTOVER_WAVES_PER_CU, 64));
      if (i > 0 && want_pipe_tail) grid = std::min<long long>(grid, (long long)al->cu_count * 8);   // (one launch's leftovers at a time: a few hundred pairs; the history slices are tens of megabytes each)
      if (full) {
        const bool h16 = b->max_len < 32000;
        const int rec = ((h16 && b->ncomp != 5) ? 2 : 4) * (band_nch[i] == 3 ? 256 : 64 * band_nch[i]);  // ints per record (2p: 16-byte entries)
        // steps of an alignment = score / g; sized for scores up to 0.9 x the read length (about 15 % divergence)
        long long records = std::max<long long>(256, (long long)(b->max_len * 0.9 * penalty_scale(b->dcfg)) / wfa::band_gcd(b->dcfg, b->ncomp == 5) + 64);
        records = knob(al, K_BAND_RECORDS, (int)records);
        band_stride[i] = ((int64_t)records * rec + 63) & ~63ll;
        const int64_t budget = free_budget(al);
        while (grid > 1 && grid * band_stride[i] * 4 > budget) grid = (grid + 1) / 2;
        need = std::max(need, (size_t)grid * band_stride[i] * 4);
        if (i == 0 && !use_fast && !use_segfull && b->max_len > knob(al, K_BAND_SPLIT_MIN, 100) && knob(al, K_BAND_NO_SPLIT, 0) == 0) {
          // split backtrace: one history slot per pair of a launch; take up to 4x the wave count (or all pairs)
          if (pb_mode) {
            pb_code_ints = ((int64_t)records * ((band_nch[i] == 3 ? 256 : 64 * band_nch[i]) / 4) + 63) & ~63ll;  // one byte per window position and step
            pb_event_ints = (((int64_t)records + 3) / 4 + 63) & ~63ll;                 // one byte per edit event (<= one per step)
            pb_stride = pb_code_ints + pb_event_ints + ((2 * (int64_t)records + 8 + 63) & ~63ll);  // + run records
          }
          const int64_t slot_bytes = (pb_mode ? pb_stride : band_stride[i]) * 4 + 16;
          // slots for every pair of the batch when they fit half of the free memory (the compact piggy-back history: C3 0.3 MB
          // per pair), otherwise for as many pairs as do (explicit history: up to 4 x the resident waves)
          int64_t pairs = in_n;
          if (pairs * slot_bytes > budget / 2) {
            // (round 6: gap-affine-2p 8 — its kernel holds 12 waves per CU, so a launch of cu_count x 64 pairs was 1.17 rounds of resident
            // waves with most of the chip idle in the second one; with twice the slots a launch is 2.7 rounds: C4-adaptive 214 -> 204 ms)
            pairs = std::min<int64_t>(in_n, (int64_t)al->cu_count * 32 * knob(al, K_BAND_SPLIT_ROUNDS, b->ncomp == 5 ? 8 : 4));
            while (pairs > 1 && pairs * slot_bytes > budget) pairs = (pairs + 1) / 2;
          }
          split_region = ((size_t)(pairs * slot_bytes) + 255) & ~(size_t)255;
          need = std::max(need, split_region);
        } else if (i > 0) {
          later_need = std::max(later_need, (size_t)grid * band_stride[i] * 4);
        }
      }
      band_grid[i] = grid;
    }
    // the band stages behind a split stage get a region of their own behind its slots when both fit: the walks of the split
    // stage's last launch then run under them
    size_t later_off = 0, gen_off = 0;
    int gen_grid = 0;
    int64_t gen_stride = g.ws_stride;
    if (split_region && later_need && (int64_t)(split_region + later_need) <= free_budget(al)) {
      later_off = split_region;
      need = std::max(need, split_region + later_need);
      if (want_pipe_tail) {   // ... and the general kernel behind them a few slices of its own
        // (a pair that reaches it has outgrown 256 diagonals: the arena of the first attempt would overflow and the pair be re-run with
        // an 8 x larger one after a host round trip — 62 ms for ONE pair of a C4-adaptive run; these few slices start that large)
        gen_grid = std::max(1, std::min(g.grid, 16));
        const size_t at = (split_region + later_need + 255) & ~(size_t)255;
        if (full) gen_stride = ((b->arena_fixed + 8 * b->arena_ints) + 63) & ~63ll;
        size_t gen_need = (size_t)gen_grid * (size_t)gen_stride * 4;
        if ((int64_t)(at + gen_need) > free_budget(al)) { gen_stride = g.ws_stride; gen_need = (size_t)gen_grid * (size_t)gen_stride * 4; }
        if ((int64_t)(at + gen_need) <= free_budget(al)) { gen_off = at; need = std::max(need, at + gen_need); }
      }
    }
    // Wide-wavefront stage (wfa_wide.hpp): exact gap-affine pairs the register windows cannot hold (or never try: reads
    // over 1.2 kb without a heuristic) — one alignment per workgroup, the wavefront rows in LDS; what it hands on goes to
    // the general kernel.  Rows as wide as LDS allows, at most the whole diagonal range of the longest pair.
    // Up to two such stages: rows in LDS (gap-affine: the faster form, 23.3 vs 21.7 k aln/s at 10 kb, but ~7 900 diagonals at
    // most), then rows in the workgroup's slice of the HBM workspace (as wide as the whole diagonal range of the longest pair: no
    // pair outgrows them; gap-affine-2p has only this form: its M ring alone is o2 + e2 + 1 rows).
    struct WideStage { wfa::WideArgs a; int grid = 0, threads = 0; size_t smem = 0, hist_off = 0; bool grows = false, w32 = false; };
    WideStage wide_stage[3];
    int n_wide = 0;
    struct TileStage { wfa::TileArgs a; int grid = 0, threads = 0; size_t smem = 0, hist_off = 0; bool on = false, w32 = false; } tile_stage;
    const bool wide_two = (b->ncomp == 5);
    if (wide_ok) {
      wfa::WideArgs w0;
      memset(&w0, 0, sizeof(w0));
      w0.g = wide_two ? wfa::band_gcd(b->dcfg, true) : wfa::gcd_int(wfa::gcd_int(b->dcfg.x, b->dcfg.o1 + b->dcfg.e1), b->dcfg.e1);
      w0.X = b->dcfg.x / w0.g; w0.OE = (b->dcfg.o1 + b->dcfg.e1) / w0.g; w0.E = b->dcfg.e1 / w0.g;
      if (wide_two) { w0.OE2 = (b->dcfg.o2 + b->dcfg.e2) / w0.g; w0.E2 = b->dcfg.e2 / w0.g; }
      w0.seq_words = ((b->max_len + 15) >> 4) + 4;
      const int nrows = wfa::wide_rows(w0.X, w0.OE, w0.E, w0.OE2, w0.E2);
      const int64_t budget = free_budget(al);
      const int full_range = (2 * b->max_len + 8 + b->dcfg.pbf + b->dcfg.tbf) & ~1;
      bool lds_covers_all = false;
      // Round 3: the pass of a step is a short dependent chain (row reads -> recurrences -> sequence reads -> stores -> barrier),
      // so a CU is kept busy by MANY pairs in flight, not by many threads on one pair: with rows in the workspace (L2) a workgroup
      // needs only its sequences in LDS, and 8 workgroups of 256 threads per CU align 10 kb reads 1.6x faster than one workgroup
      // of 1 024 threads with its rows in LDS (36.8 k vs 23.3 k aln/s; gap-affine-2p: 4 x 512 threads, +25 %).  The LDS form
      // (north_star's layout) remains what a small batch gets: all of a CU's threads on one pair is the shortest latency.
      // WFA_HIP_WIDE_GROWS: 1 = workspace rows always, 0 = LDS rows first always
      const int grows_knob = knob(al, K_WIDE_GROWS, -1);
      const bool many_pairs = (int64_t)in_n >= (int64_t)al->cu_count * 2;
      const bool prefer_ws = grows_knob >= 0 ? grows_knob != 0 : many_pairs;
      // threads per workgroup of the workspace form: 256 (gap-affine-2p: 512) once the batch has a pair for every CU's four
      // workgroup slots of that size — measured on 2 000 x 10 kb: 128 threads 35.1 k, 256: 36.8 k, 384: 31.0 k, 512: 32.6 k, 1 024: 21.6 k aln/s
      // (8 192 x 10 kb: 128 threads 200 / 231 ms score / full, 256: 207 / 241; C4 as written, 4 096 pairs: 256 threads 705 ms, 512: 740,
      // 384: 925, 128: 719; the 2p form compiled for six waves per SIMD — three 512-thread workgroups per CU instead of two — 1 047 ms:
      // more workgroups in flight than the rows' L2 footprint allows cost more than the latency they hide)
      int ws_threads = 1024;
      if (!wide_two && (int64_t)in_n >= (int64_t)al->cu_count * 16) ws_threads = 128;
      else if ((int64_t)in_n >= (int64_t)al->cu_count * 8) ws_threads = 256;
      else if ((int64_t)in_n >= (int64_t)al->cu_count * 4) ws_threads = wide_two ? 512 : 256;
      else if ((int64_t)in_n >= (int64_t)al->cu_count * 2) ws_threads = 512;
      // Round 4: the temporally blocked form (wfa_tile.hpp) goes first for exact alignments with int16 rows: a wave advances a
      // block of diagonals by T steps inside an LDS tile, the rows in the workspace are read and written once per T steps.  What it
      // hands on (pairs where the reference's per-step trimming would change a value, a history that does not fit) goes to the
      // step-by-step forms below.  WFA_HIP_TILE=0: off; WFA_HIP_TILE_T / _WT / _THREADS / _PER_CU: geometry
      // (round 6: reads of up to 32 000 bases on int16 rows — their NULL is -32768.  Longer ones: wfa_tile_kernel<.., W32> — int32 cells,
      // the sequences read from global memory (50 KB of them per workgroup in LDS left one workgroup per CU), eight waves per pair.  One
      // workgroup per pair needs pairs to fill the chip: exact 100 kb score, 512 / 1 024 pairs 345 / 351 aln/s against 275 / 192 on the
      // step-by-step forms below, but 256 pairs 238 against 290 and 64 pairs 62 against 104 — so from two pairs per CU on.
      // WFA_HIP_TILE32 = 0 never, 1 always)
      const bool tile32 = b->max_len > WFA_TILE_MAX_LEN;
      const int tile_cb = tile32 ? 4 : 2;
      const int tile32_knob = knob(al, K_TILE32, -1);
      const bool tile32_on = tile32_knob >= 0 ? tile32_knob != 0 : (int64_t)in_n >= 2 * (int64_t)al->cu_count;
      if (b->dcfg.heuristic == WFA_HEUR_NONE && (!tile32 || (tile32_on && 2 * (int64_t)b->max_len < ((int64_t)1 << 29))) && knob(al, K_TILE, 1) != 0) {
        wfa::TileArgs& ta = tile_stage.a;
        memset(&ta, 0, sizeof(ta));
        wfa::TileGeom& tg = ta.g;
        tg.X = w0.X; tg.OE = w0.OE; tg.E = w0.E; tg.OE2 = w0.OE2; tg.E2 = w0.E2;
        tg.DM = std::max(std::max(tg.X, tg.OE), tg.OE2);
        // geometry by read length (round 5): reads of up to 8 kb have wavefronts of a few hundred diagonals and scores of a few hundred
        // steps — tiles of 128 diagonals advanced 8 steps by two waves per alignment keep more alignments in flight and waste less of
        // a tile on halo (1 kb at 5 %: 4.3 -> 6.6 M aln/s, 2 kb at 1 %: 6.8 -> 15.1 M, 600 bp at 10 %: 3.9 -> 5.9 M; 4 kb at 5 %: equal)
        const bool small_geom = !wide_two && b->max_len <= 8000;   // (3 / 5 / 7 kb at 5 %: equal / +10 % / +2 % score, 5 kb full +35 %; 10 kb: -20 %)
        tile_stage.w32 = tile32;
        tg.T = knob(al, K_TILE_T, (wide_two || small_geom) ? 8 : 16) & ~1;
        tg.Wt = tile32 ? (wide_two ? 128 : 256) : knob(al, K_TILE_WT, (wide_two || small_geom) ? 128 : 256);   // (int32 cells: the two widths that exist)
        const int bw = tg.Wt - 2 * tg.T;
        // (gap-affine-2p reads of up to 1.2 kb: one wave per alignment — 300 / 600 / 1 000 bp at 8 %: +28 / +51 / +29 % over four;
        // round 6: gap-affine reads over 16 kb eight waves — 30 kb, 512 pairs: 128 / 256 / 512 threads 4.6 / 8.7 / 11.7 k aln/s; 10 kb keeps 256)
        const int tthreads = std::max(64, std::min(512, knob(al, K_TILE_THREADS, (tile32 || (!wide_two && b->max_len > 16000)) ? 512 : small_geom ? 128 : (wide_two && b->max_len <= 1200) ? 64 : 256) & ~63));
        if (tg.T >= 2 && tg.T <= WFA_TILE_MAX_T && tg.Wt >= 64 && tg.Wt % 64 == 0 && tg.Wt <= 256 && bw >= 16 &&
            wfa::tile_cand_count(tg) <= WFA_TILE_MAX_ROWS) {
          ta.gs = w0.g; ta.seq_words = tile32 ? 0 : w0.seq_words;   // (int32 form: the sequences stay in global memory)
          const int nbmax = (2 * b->max_len + 1 + bw - 1) / bw;
          ta.rwh = (nbmax * bw + 2 * tg.T + 1) & ~1;
          ta.rows_stride = ((int64_t)wfa::tile_hbm_rows(tg) * ta.rwh + 63) & ~63ll;
          tile_stage.threads = tthreads;
          tile_stage.smem = wfa::tile_smem_bytes(tg, ta.seq_words, tthreads / 64, tile_cb);
          if (tile_stage.smem <= (size_t)160 * 1024) {
            int per_cu = wfa::tile_occupancy(full, wide_two, tg.Wt / 64, tthreads, tile_stage.smem, tile32);
            per_cu = std::max(1, std::min(per_cu, knob(al, K_TILE_PER_CU, 16)));
            tile_stage.grid = (int)std::min<int64_t>((int64_t)al->cu_count * per_cu, in_n);
            int64_t hist_bytes = 0;
            if (full) hist_bytes = (int64_t)full_range * ((int64_t)(b->max_len * 0.9 * penalty_scale(b->dcfg)) / w0.g + 64 + tg.T) / 2 + (1 << 20);
            while (tile_stage.grid > 1 && (int64_t)tile_stage.grid * (ta.rows_stride * tile_cb + hist_bytes) > budget) tile_stage.grid = (tile_stage.grid + 1) / 2;
            if ((int64_t)tile_stage.grid * (ta.rows_stride * tile_cb + hist_bytes) <= budget) {
              ta.hist_stride = (hist_bytes / 4) & ~15ll;
              tile_stage.hist_off = ((size_t)tile_stage.grid * (size_t)ta.rows_stride * tile_cb + 255) & ~(size_t)255;
              need = std::max(need, tile_stage.hist_off + (size_t)tile_stage.grid * (size_t)ta.hist_stride * 4);
              tile_stage.on = true;
            }
          }
        }
      }
      if (!wide_two && !wide32 && !prefer_ws && !wide_adapt) {
        WideStage& st = wide_stage[n_wide];
        st.a = w0;
        const size_t lds_max = (size_t)std::min(160, std::max(16, knob(al, K_WIDE_LDS_KB, 160))) * 1024;
        const size_t fixed = wfa::wide_smem_bytes(w0.X, w0.OE, w0.E, 0, 0, 0, w0.seq_words, true);
        if (fixed + (size_t)nrows * 2 * 512 <= lds_max) {
          int wcap = (int)((lds_max - fixed) / ((size_t)nrows * 2)) - 4;
          wcap = std::min(wcap, full_range) & ~1;
          lds_covers_all = wcap >= full_range;
          st.a.wcap = wcap;
          st.smem = wfa::wide_smem_bytes(w0.X, w0.OE, w0.E, 0, 0, wcap, w0.seq_words, true);
          const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, ((size_t)160 * 1024) / st.smem));
          st.threads = knob(al, K_WIDE_THREADS, per_cu >= 4 ? 256 : (per_cu >= 2 ? 512 : 1024));
          st.grid = (int)std::min<int64_t>((int64_t)al->cu_count * per_cu, in_n);
          if (full) {
            // history: one byte per cell up to the step where a wavefront would outgrow the rows, + directory + events
            const int64_t tmax = (int64_t)wcap * w0.E / 2 + 64;
            int64_t bytes = (int64_t)wcap * wcap * w0.E / 4 + 16 * tmax + (1 << 16);
            while (st.grid > 1 && (int64_t)st.grid * bytes > budget) st.grid = (st.grid + 1) / 2;
            bytes = std::min<int64_t>(bytes, budget / std::max(st.grid, 1));
            st.a.hist_stride = (bytes / 4) & ~15ll;
            need = std::max(need, (size_t)st.grid * (size_t)st.a.hist_stride * 4);
          }
          ++n_wide;
        }
      }
      // Round 4: what the banded stages hand on under wf-adaptive (47 of 8 192 pairs of 100 kb: a wavefront beyond their 256 diagonals)
      // is a few hundred diagonals wide and a handful of pairs — one alignment's latency is all that counts.  Rows in LDS (int32 beyond
      // 16 kb), as many diagonals as fit beside the sequences, one workgroup per CU: a step costs LDS round trips instead of L2 ones
      // (100 kb: 5.2 us per step there).  A wavefront that leaves the rows goes on to the workspace form below.
      if (wide_adapt && knob(al, K_WIDE_ADAPT_LDS, 1) != 0) {
        WideStage& st = wide_stage[n_wide];
        st.a = w0; st.grows = false; st.w32 = wide32;
        const int ob = wide32 ? 4 : 2;
        const size_t lds_max = (size_t)std::min(160, std::max(16, knob(al, K_WIDE_LDS_KB, 160))) * 1024;
        const size_t fixed = wfa::wide_smem_bytes(w0.X, w0.OE, w0.E, w0.OE2, w0.E2, 0, w0.seq_words, true, ob);
        if (fixed + (size_t)nrows * ob * 516 <= lds_max) {
          int wcap = (int)((lds_max - fixed) / ((size_t)nrows * ob)) - 4;
          wcap = std::min(std::min(wcap, 4096), full_range) & ~1;
          if (knob(al, K_WIDE_ADAPT_LDS, 1) > 1) wcap = std::min(wcap, knob(al, K_WIDE_ADAPT_LDS, 1) & ~1);   // (tests: rows so narrow that pairs go on to the workspace form)
          st.a.wcap = wcap;
          st.smem = wfa::wide_smem_bytes(w0.X, w0.OE, w0.E, w0.OE2, w0.E2, wcap, w0.seq_words, true, ob);
          st.threads = knob(al, K_WIDE_THREADS, 512);
          st.grid = (int)std::min<int64_t>((int64_t)al->cu_count, in_n);
          bool fits = true;
          if (full) {
            int64_t hist_bytes = std::min<int64_t>((int64_t)full_range * ((int64_t)(b->max_len * 0.9 * penalty_scale(b->dcfg)) / w0.g + 64) / 2 + (1 << 20),
                                                   ((int64_t)(b->max_len * 1.2) / w0.g + 64) * 2048 + (1 << 20));
            while (st.grid > 1 && (int64_t)st.grid * hist_bytes > budget) st.grid = (st.grid + 1) / 2;
            fits = (int64_t)st.grid * hist_bytes <= budget;
            st.a.hist_stride = (hist_bytes / 4) & ~15ll;
            if (fits) need = std::max(need, (size_t)st.grid * (size_t)st.a.hist_stride * 4);
          }
          if (fits) ++n_wide;
        }
      }
      if (!lds_covers_all) {
        WideStage& st = wide_stage[n_wide];
        st.a = w0; st.grows = true; st.w32 = wide32;
        st.a.wcap = full_range;
        st.a.rows_stride = (int64_t)((nrows * wfa::wide_row_halfs(full_range) + 63) & ~(size_t)63);
        const int64_t row_bytes = wide32 ? 4 : 2;   // bytes per offset
        st.smem = wfa::wide_smem_bytes(w0.X, w0.OE, w0.E, w0.OE2, w0.E2, full_range, w0.seq_words, false);
        const bool seqs_fit_lds = st.smem <= (size_t)160 * 1024;   // (both packed sequences are staged in LDS: reads up to ~300 kb)
        st.threads = knob(al, K_WIDE_THREADS, wide_adapt ? 512 : ws_threads);   // (wf-adaptive: a cut wavefront is a few hundred diagonals: one cell per thread)
        st.grid = (int)std::min<int64_t>((int64_t)al->cu_count * std::max<int64_t>(1, std::min<int64_t>(2048 / st.threads, (160 * 1024) / std::max<size_t>(st.smem, 1))), in_n);
        if (wide_adapt) st.grid = std::min(st.grid, al->cu_count);   // (leftovers: one workgroup per CU is plenty)
        int64_t hist_bytes = 0;
        if (full) hist_bytes = (int64_t)full_range * ((int64_t)(b->max_len * 0.9 * penalty_scale(b->dcfg)) / w0.g + 64) / 2 + (1 << 20);   // one byte per cell, + directory + events
        // (wf-adaptive: the cut-off keeps the wavefronts narrow: ~2 KB of codes per step; a pair that needs more is handed on)
        if (full && wide_adapt) hist_bytes = std::min<int64_t>(hist_bytes, ((int64_t)(b->max_len * 1.2) / w0.g + 64) * 2048 + (1 << 20));
        while (st.grid > 1 && (int64_t)st.grid * (st.a.rows_stride * row_bytes + hist_bytes) > budget) st.grid = (st.grid + 1) / 2;
        if (seqs_fit_lds && (int64_t)st.grid * (st.a.rows_stride * row_bytes + hist_bytes) <= budget) {
          st.a.hist_stride = (hist_bytes / 4) & ~15ll;
          st.hist_off = ((size_t)st.grid * (size_t)st.a.rows_stride * (size_t)row_bytes + 255) & ~(size_t)255;
          need = std::max(need, st.hist_off + (size_t)st.grid * (size_t)st.a.hist_stride * 4);
          ++n_wide;
        }
      }
    }
    // full CIGARs of short reads: 16-lane segments over the whole batch (history slot per pair, several launches), then
    // 32- and 64-lane segments over what was handed on (device-side counts: slots for 1/8 resp. 1/32 of the batch)
    int64_t segfull_slot_ints[3] = {0, 0, 0}, segfull_cap[3] = {0, 0, 0};
    int segfull_w[3] = {16, 32, 64};
    int n_segfull = use_segfull ? 3 : 0;
    if (use_segfull && b->stage_pick != 0 && !al->knobs.set[K_SEGFULL_STAGES]) {   // (the pilot of batch_build chose the first width)
      n_segfull = 0;
      for (int w = 16; w <= 64; w *= 2) if (w >= b->stage_pick) segfull_w[n_segfull++] = w;
    } else if (use_segfull) {
      n_segfull = std::max(0, std::min(3, knob(al, K_SEGFULL_STAGES, 3)));
    }
    // Round 3: the 16-diagonal stage of that cascade is the lane-per-pair kernel with origin codes (wfa_lane_kernel<.., FULL>):
    // it walks in-kernel and leaves only run records in HBM (WFA_LANE_RUN_SLOT ints + an end state per pair), so the whole batch
    // is one launch; WFA_HIP_LANE_FULL=0 keeps the 16-lane segments (round 4: with piggy-back code records) as the first stage
    int lfs_x = 0, lfs_oe = 0, lfs_e = 0;
    const bool lane_shape_fits = wfa::seg_shape(b->dcfg, &lfs_x, &lfs_oe, &lfs_e) != WFA_SHAPE_RTC || wfa::rtc_lane_shape_ok(lfs_x, lfs_oe, lfs_e);
    const bool use_lanefull = use_segfull && n_segfull >= 1 && segfull_w[0] == 16 && knob(al, K_LANE_FULL, 1) != 0 && lane_shape_fits;
    // (behind it one segment stage, 32 lanes, then the banded kernel: the 64-lane stage cost more in launches than its ~300 pairs per
    // million are worth)
    if (use_lanefull) { for (int i = 1; i < n_segfull; ++i) segfull_w[i - 1] = segfull_w[i]; --n_segfull; if (!al->knobs.set[K_SEGFULL_STAGES]) n_segfull = std::min(n_segfull, 1); }
    if (use_segfull) {
      for (int i = 0; i < n_segfull; ++i) {
        const int rank = i + (use_lanefull ? 1 : 0);   // position in the cascade: 0 = takes the whole batch
        { long long ci, ei, si; wfa::seg_full_slot(b->dcfg, segfull_w[i], b->max_len, &ci, &ei, &si); segfull_slot_ints[i] = si; }   // code records of w bytes, events, runs
        const int64_t slot_bytes = segfull_slot_ints[i] * 4 + (int64_t)sizeof(int4);
        int64_t want = (rank == 0) ? std::min<int64_t>((int64_t)knob(al, K_SEGFULL_PAIRS, 2000000), std::max<int64_t>(1, ((int64_t)8 << 30) / slot_bytes))
                                   : std::max<int64_t>(4096, (int64_t)in_n / (rank == 1 ? 8 : 32));
        // (the stage that takes the whole batch runs it as two balanced launches at least, once each still fills the chip: the walks and
        // the expand of a launch — a third of the stage's time at 10 % divergence — run under the alignment kernel of the next)
        if (rank == 0 && in_count == nullptr && (int64_t)in_n >= 262144 && !al->knobs.set[K_SEGFULL_PAIRS] && knob(al, K_NO_DUAL, 0) == 0)
          want = std::min<int64_t>(want, (((int64_t)in_n + 1) / 2 + 63) & ~63ll);
        segfull_cap[i] = std::max<int64_t>(1, std::min<int64_t>(in_n, std::min<int64_t>(want, free_budget(al) / slot_bytes)));
        // (several launches: two slot arrays, the walks of a launch run under the alignment kernel of the next)
        need = std::max(need, (size_t)(segfull_cap[i] * slot_bytes) * ((rank == 0 && segfull_cap[i] < (int64_t)in_n) ? 2 : 1));
      }
    }
    // the lane-full stage's region lies BEHIND everything the later stages use (its expand runs on the side stream under them)
    const int64_t lanefull_slot_bytes = (int64_t)WFA_LANE_RUN_SLOT * 4 + (int64_t)sizeof(int4);
    int64_t lanefull_cap = 0;
    long long lanefull_grid = 0;
    int lanefull_recs = 0, lf_x = 0, lf_oe = 0, lf_e = 0, lf_min_pairs = 256;
    size_t lanefull_off = 0, lanefull_codes_off = 0, lanefull_codes_bytes = 0, lanefull_list_off = 0;
    int lanefull_split = 1;   // > 1: every launch has slots and code lists of its own
    if (use_lanefull) {
      (void)wfa::seg_shape(b->dcfg, &lf_x, &lf_oe, &lf_e);
      lanefull_off = (need + 255) & ~(size_t)255;
      // per launch: run records + end state per pair, and the waves' record lists of comparison bits (512 bytes per wave-step)
      const int64_t lf_budget = std::max<int64_t>((int64_t)free_budget(al) - (int64_t)lanefull_off, (int64_t)1 << 20);
      lanefull_cap = std::max<int64_t>(1, std::min<int64_t>(in_n, lf_budget / (lanefull_slot_bytes + 256)));
      if (al->knobs.set[K_SEGFULL_PAIRS]) lanefull_cap = std::min<int64_t>(lanefull_cap, std::max<int64_t>(64, knob(al, K_SEGFULL_PAIRS, 2000000)));   // (tests: several launches over one region)
      // When the whole batch fits (the usual case) it is cut into balanced launches with code lists of their own: walk + expand of a
      // launch (latency- and memory-bound) run on the side stream under the lane kernel (issue-bound) of the next
      if (lanefull_cap >= (int64_t)in_n) {
        lanefull_split = std::max(1, std::min(8, knob(al, K_LANE_FULL_SPLIT, (int64_t)in_n >= 524288 ? 2 : 1)));
        lanefull_cap = ((((int64_t)in_n + lanefull_split - 1) / lanefull_split) + 63) & ~63ll;
      }
      // pairs per wave: 256 for launches that fill the chip; a launch of up to 256 k pairs is cut into lane-fulls of 64 (a run is as long as
      // a wave's life: 65 536 pairs with CIGAR 397 -> 289 us; 1 M pairs in two launches: 1.07 ms with 256, 1.15 with 64).  One value for the
      // geometry and every launch of the stage: a shorter last launch must not have more waves than the record lists were sized for
      lf_min_pairs = knob(al, K_LANE_MIN_PAIRS, lanefull_cap <= 262144 ? 64 : 256);
      wfa::lane_full_geometry((uint32_t)lanefull_cap, al->cu_count, knob(al, K_LANE_WAVES_PER_CU, 48), lf_min_pairs, lf_oe, lf_e,
                              &lanefull_grid, &lanefull_recs);
      lanefull_codes_off = lanefull_off + (((size_t)(lanefull_cap * lanefull_split * lanefull_slot_bytes) + 255) & ~(size_t)255);
      lanefull_codes_bytes = ((size_t)lanefull_grid * (size_t)lanefull_recs * 512 + 255) & ~(size_t)255;
      lanefull_list_off = lanefull_codes_off + lanefull_codes_bytes * (size_t)lanefull_split;
      need = std::max(need, lanefull_list_off + (size_t)in_n * sizeof(uint32_t));
    }
    int rc = ensure_ws(al, need);
    if (rc != WFA_HIP_OK) return rc;

    DualStream lane_expands{al, stream, 2};   // the expand of the lane-full stage, left running under the stages behind it
    if (use_lanefull) {
      uint32_t* out_list = b->d_fb_list2[out_sel];
      uint32_t* out_count = next_count();
      int X, OE, E;
      const int shape = wfa::seg_shape(b->dcfg, &X, &OE, &E);
      wfa::FastArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.lin = b->dcfg.lin;
      fa.words = b->d_words; fa.meta = b->d_meta; fa.worklist = in_list; fa.nwork_dev = nullptr;
      fa.score = b->d_score; fa.status = b->d_status; fa.fb_list = out_list; fa.fb_count = out_count;
      fa.g = wfa::gcd_int(wfa::gcd_int(b->dcfg.x, b->dcfg.o1 + b->dcfg.e1), b->dcfg.e1);
      const int64_t lf_slots = lanefull_cap * lanefull_split;   // slots in the region (split: slot = work item; otherwise one launch's worth, reused)
      int32_t* const lf_runs = al->ws + lanefull_off / 4;
      int4* const lf_ends = reinterpret_cast<int4*>(reinterpret_cast<char*>(al->ws) + lanefull_off + (size_t)lf_slots * WFA_LANE_RUN_SLOT * 4);
      fa.hist = lf_runs; fa.hist_stride = WFA_LANE_RUN_SLOT; fa.end_state = lf_ends;
      fa.codes = reinterpret_cast<uint2*>(reinterpret_cast<char*>(al->ws) + lanefull_codes_off); fa.codes_cap = lanefull_recs;
      wfa::BandArgs ba;
      memset(&ba, 0, sizeof(ba));
      ba.meta = b->d_meta; ba.worklist = in_list; ba.words = b->d_words;
      ba.cigar_ops = b->d_ops; ba.cigar_off = b->d_cigar_off; ba.cigar_begin = b->d_cigar_begin; ba.cigar_len = b->d_cigar_len;
      ba.hist = fa.hist; ba.hist_stride = fa.hist_stride; ba.end_state = fa.end_state;
      // (the walk hands a pair with more runs than its slot holds to a list of its own, taken by the general kernel after the join:
      // walk and expand both run beside the later stages)
      uint32_t* const walk_list = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(al->ws) + lanefull_list_off);
      ba.lane_codes = fa.codes; ba.status = b->d_status; ba.fb_list = walk_list; ba.fb_count = b->d_counters + 8;
      ba.g = fa.g; ba.x = b->dcfg.x; ba.oe = b->dcfg.o1 + b->dcfg.e1; ba.e = b->dcfg.e1;
      // one launch (the usual case): the op bytes are written on the side stream while the main stream goes on with what was handed on
      { const int drc = lane_expands.begin(lf_slots >= (int64_t)in_n); if (drc != WFA_HIP_OK) return drc; }
      int64_t launch = 0;
      for (int64_t w0 = 0; w0 < (int64_t)in_n; w0 += lanefull_cap, ++launch) {
        const uint32_t cnt = (uint32_t)std::min<int64_t>(lanefull_cap, (int64_t)in_n - w0);
        fa.work_begin = (uint32_t)w0; fa.nwork = cnt;
        if (lf_slots >= (int64_t)in_n) {   // slots and code lists of this launch
          fa.hist = lf_runs + w0 * WFA_LANE_RUN_SLOT; fa.end_state = lf_ends + w0;
          fa.codes = reinterpret_cast<uint2*>(reinterpret_cast<char*>(al->ws) + lanefull_codes_off + (size_t)launch * lanefull_codes_bytes);
          ba.hist = fa.hist; ba.end_state = fa.end_state; ba.lane_codes = fa.codes;
        }
        // (a shorter last launch has fewer waves and needs fewer records than the lists were sized for)
        if (wfa::launch_lane_args(shape, OE, E, al->cu_count, knob(al, K_LANE_WAVES_PER_CU, 48), knob(al, K_LANE_REFILL_MIN, 8), b->max_len, stream, fa, true, 0, lf_min_pairs, false, X) != 0) {
          al->err = "lane kernel launch failed"; return WFA_HIP_EDEVICE;
        }
        ba.work_begin = (uint32_t)w0; ba.nwork = cnt;
        hipStream_t es_ = stream;
        { const int drc = lane_expands.walk_stream(launch, &es_); if (drc != WFA_HIP_OK) return drc; }
        if (wfa::launch_lane_expand(ba, es_, es_) != 0) { al->err = "walk / expand launch failed"; return WFA_HIP_EDEVICE; }
        { const int drc = lane_expands.after_walk(launch); if (drc != WFA_HIP_OK) return drc; }
      }
      if (first_stage) b->last_kernel_pairs = in_n;
      in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
    }
    for (int sf = 0; sf < n_segfull; ++sf) {
      uint32_t* out_list = b->d_fb_list2[out_sel];
      uint32_t* out_count = next_count();
      wfa::FastArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.lin = b->dcfg.lin;
      fa.words = b->d_words; fa.meta = b->d_meta; fa.worklist = in_list; fa.nwork_dev = in_count;
      fa.score = b->d_score; fa.status = b->d_status; fa.fb_list = out_list; fa.fb_count = out_count;
      fa.hist = al->ws; fa.hist_stride = segfull_slot_ints[sf];
      fa.end_state = reinterpret_cast<int4*>(reinterpret_cast<char*>(al->ws) + (size_t)segfull_cap[sf] * segfull_slot_ints[sf] * 4);
      wfa::BandArgs ba;
      memset(&ba, 0, sizeof(ba));
      ba.meta = b->d_meta; ba.worklist = in_list; ba.words = b->d_words;
      ba.cigar_ops = b->d_ops; ba.cigar_off = b->d_cigar_off; ba.cigar_begin = b->d_cigar_begin; ba.cigar_len = b->d_cigar_len;
      ba.g = wfa::gcd_int(wfa::gcd_int(b->dcfg.x, b->dcfg.o1 + b->dcfg.e1), b->dcfg.e1);
      ba.x = b->dcfg.x; ba.oe = b->dcfg.o1 + b->dcfg.e1; ba.e = b->dcfg.e1;
      ba.hist = al->ws; ba.hist_stride = segfull_slot_ints[sf]; ba.end_state = fa.end_state;
      ba.split = 1; ba.h16 = 1; ba.seg_w = segfull_w[sf];
      { long long ci, ei, si; wfa::seg_full_slot(b->dcfg, segfull_w[sf], b->max_len, &ci, &ei, &si); ba.pb = 1; ba.pb_code_ints = ci; ba.pb_event_ints = ei; }
      // (first stage: the host knows the count and walks it in launches of `cap` pairs; later stages: one launch over
      // the device-side list, slots for `cap` of its pairs)
      const int64_t total = (in_count == nullptr) ? (int64_t)in_n : segfull_cap[sf];
      const size_t half_bytes = ((size_t)segfull_cap[sf] * (size_t)(segfull_slot_ints[sf] * 4 + (int64_t)sizeof(int4)) + 255) & ~(size_t)255;
      DualStream dual{al, stream};
      { const int drc = dual.begin(sf == 0 && in_count == nullptr && total > segfull_cap[sf] && al->ws_bytes >= 2 * half_bytes); if (drc != WFA_HIP_OK) return drc; }
      int64_t launch = 0;
      for (int64_t w0 = 0; w0 < total; w0 += segfull_cap[sf], ++launch) {
        const uint32_t cnt = (uint32_t)std::min<int64_t>(segfull_cap[sf], total - w0);
        char* base = reinterpret_cast<char*>(al->ws) + ((dual.on && (launch & 1)) ? half_bytes : 0);
        fa.hist = reinterpret_cast<int32_t*>(base);
        fa.end_state = reinterpret_cast<int4*>(base + (size_t)segfull_cap[sf] * segfull_slot_ints[sf] * 4);
        ba.hist = fa.hist; ba.end_state = fa.end_state;
        fa.work_begin = (uint32_t)w0; fa.nwork = cnt;
        { const int drc = dual.before_align(launch); if (drc != WFA_HIP_OK) return drc; }
        if (in_count != nullptr) HIP_TRY(al, hipMemsetAsync(fa.end_state, 0, (size_t)cnt * sizeof(int4), stream));
        if (wfa::launch_seg_full(b->dcfg, al->cu_count, knob(al, K_FAST_WAVES_PER_CU, 256), stream, fa, segfull_w[sf]) != 0) { al->err = "segmented kernel launch failed"; return WFA_HIP_EDEVICE; }
        ba.work_begin = (uint32_t)w0; ba.nwork = cnt;
        hipStream_t ws_ = stream;
        { const int drc = dual.walk_stream(launch, &ws_); if (drc != WFA_HIP_OK) return drc; }
        if (wfa::launch_band_bt(ba, 1, ws_) != 0) { al->err = "backtrace launch failed"; return WFA_HIP_EDEVICE; }
        { const int drc = dual.after_walk(launch); if (drc != WFA_HIP_OK) return drc; }
      }
      { const int drc = dual.end(); if (drc != WFA_HIP_OK) return drc; }
      if (first_stage) b->last_kernel_pairs = in_n;
      in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
    }
    if (use_fast) {
      // register-kernel stages, each taking what the one before handed on (WFA_HIP_FAST_STAGES, one digit per
      // stage): 1 = the lane-per-pair kernel (64 pairs per wave, band of 16 diagonals, wfa_lane.hpp); 7/6/8/9 = segments
      // of 8/16/32/64 lanes with the two-round (lazy) extension, 3/2/4/5 = the same widths extending every cell at once
      const char* stages_env = al->knobs.fast_stages.empty() ? nullptr : al->knobs.fast_stages.c_str();
      const char* stages = stages_env ? stages_env : "189";
      if (!stages_env && b->stage_pick != 0 && in_count == nullptr) {   // (the pilot of batch_build chose the first width)
        stages = (b->stage_pick == 16) ? "189" : (b->stage_pick == 32) ? "89" : "9";  // (128: 64 lanes still take the pairs that fit)
      }
      int variants[6] = {-1, -1, -1, -1, -1, -1};
      int nv = 0;
      for (const char* c = stages; *c && nv < 6; ++c) {
        const int v = *c - '0';
        if (v < 0 || v > 9) continue;
        if (v < 1) continue;
        if (v == 1) {   // (the lane kernel of a run-time shape: only while its rings fit the register file)
          int X_, OE_, E_;
          if (wfa::seg_shape(b->dcfg, &X_, &OE_, &E_) == WFA_SHAPE_RTC && !wfa::rtc_lane_shape_ok(X_, OE_, E_)) continue;
        }
        variants[nv++] = v;
      }
      if (nv == 0) variants[nv++] = 6;
      for (int pass = 0; pass < nv; ++pass) {
        uint32_t* out_list = b->d_fb_list2[out_sel];
        uint32_t* out_count = next_count();
        hipEvent_t se0 = nullptr, se1 = nullptr;
        const bool stage_timing = knob(al, K_STAGE_TIMING, 0) != 0;
        if (stage_timing) { hipEventCreate(&se0); hipEventCreate(&se1); hipEventRecord(se0, stream); }
        int lrc;
        if (variants[pass] == 1) {
          int X, OE, E;
          const int shape = wfa::seg_shape(b->dcfg, &X, &OE, &E);
          // (slices of the list taken at run time: as many waves as the chip holds at once — every wave goes on until the list is used up)
          const uint32_t lane_dyn = (in_count == nullptr && in_n >= 65536u) ? (uint32_t)std::max(0, knob(al, K_LANE_DYN, 192)) : 0u;
          lrc = wfa::launch_lane(shape, wfa::gcd_int(wfa::gcd_int(b->dcfg.x, b->dcfg.o1 + b->dcfg.e1), b->dcfg.e1), al->cu_count,
                                 lane_dyn ? knob(al, K_LANE_DYN_WAVES, 16) : knob(al, K_LANE_WAVES_PER_CU, 48), knob(al, K_LANE_REFILL_MIN, 8), b->max_len, stream, b->d_words, b->d_meta,
                                 in_list, in_count, in_n, b->d_score, b->d_status, out_list, out_count,
                                 (knob(al, K_LANE_DEBUG, 0) && al->ws) ? al->ws : nullptr, knob(al, K_LANE_LDS_PAD_KB, 0), X, OE, E, knob(al, K_LANE_MIN_PAIRS, 0),
                                 // (slices of the list taken at run time, 256 pairs at a time, from a counter zeroed with the run's others;
                                 // a list whose length only the device knows keeps the fixed slices; WFA_HIP_LANE_DYN=0: fixed slices)
                                 b->d_counters + 12, lane_dyn);
          if (knob(al, K_LANE_DEBUG, 0) && al->ws) {  // development aid (build with -DWFA_LANE_DEBUG_COUNTERS=1)
            unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            hipStreamSynchronize(stream); hipMemcpy(c, al->ws, sizeof(c), hipMemcpyDeviceToHost); hipMemset(al->ws, 0, sizeof(c));
            fprintf(stderr, "[wfa_hip] lane kernel: %llu wave-steps, %llu refills, %llu parked runs, %llu parked rounds, %llu probe blocks, %llu second-run rounds, "
                            "%llu hand-over blocks, %llu second runs\n", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
          }
        } else {
          lrc = wfa::launch_seg(b->dcfg, al->cu_count, knob(al, K_FAST_WAVES_PER_CU, 256), stream, b->d_words, b->d_meta, in_list, in_count,
                                in_n, b->d_score, b->d_status, out_list, out_count, variants[pass]);
        }
        if (lrc != 0) { al->err = "fast kernel launch failed"; return WFA_HIP_EDEVICE; }
        if (stage_timing) {  // development aid: synchronises after every stage
          hipEventRecord(se1, stream); hipEventSynchronize(se1);
          float ms = 0.f; hipEventElapsedTime(&ms, se0, se1);
          uint32_t handed = 0; hipMemcpy(&handed, out_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
          fprintf(stderr, "[wfa_hip] stage %d (variant %d): %.3f ms, handed on %u pairs\n", pass, variants[pass], ms, handed);
          hipEventDestroy(se0); hipEventDestroy(se1);
        }
        if (first_stage) b->last_kernel_pairs = in_n;
        in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
      }
    }
    if (use_laneh) {
      uint32_t* out_list = b->d_fb_list2[out_sel];
      uint32_t* out_count = next_count();
      wfa::FastArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.words = b->d_words; fa.meta = b->d_meta; fa.worklist = in_list; fa.nwork_dev = in_count; fa.nwork = in_n;
      fa.score = b->d_score; fa.status = b->d_status; fa.fb_list = out_list; fa.fb_count = out_count;
      fa.g = wfa::gcd_int(wfa::gcd_int(b->dcfg.x, b->dcfg.o1 + b->dcfg.e1), b->dcfg.e1);
      fa.ef = (b->dcfg.endsfree && (b->dcfg.pbf | b->dcfg.pef | b->dcfg.tbf | b->dcfg.tef)) ? 1 : 0;
      fa.pbf = b->dcfg.pbf; fa.pef = b->dcfg.pef; fa.tbf = b->dcfg.tbf; fa.tef = b->dcfg.tef;
      fa.heur = b->dcfg.heuristic; fa.min_wf_len = b->dcfg.min_wf_len; fa.max_dist_thr = b->dcfg.max_dist_thr;
      fa.steps_between = b->dcfg.steps_between; fa.max_steps = b->dcfg.max_steps;
      const int shape = wfa::seg_shape(b->dcfg, &lh_x, &lh_oe, &lh_e);
      // (slices of the list taken at run time, as the plain score-only form)
      const uint32_t laneh_dyn = (in_count == nullptr && in_n >= 65536u) ? (uint32_t)std::max(0, knob(al, K_LANE_DYN, 192)) : 0u;
      if (laneh_dyn) { fa.dyn_next = b->d_counters + 13; fa.dyn_chunk = laneh_dyn; }
      if (wfa::launch_lane_args(shape, lh_oe, lh_e, al->cu_count, laneh_dyn ? knob(al, K_LANE_DYN_WAVES, 16) : knob(al, K_LANE_WAVES_PER_CU, 48), knob(al, K_LANE_REFILL_MIN, 8), b->max_len, stream, fa, false, 0, knob(al, K_LANE_MIN_PAIRS, 0), laneh_form, lh_x) != 0) {   // (pairs per wave by the size of the batch, as the plain form)
        al->err = "lane kernel launch failed"; return WFA_HIP_EDEVICE;
      }
      if (first_stage) b->last_kernel_pairs = in_n;
      in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
    }
    if (use_segh) {
      uint32_t* out_list = b->d_fb_list2[out_sel];
      uint32_t* out_count = next_count();
      wfa::FastArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.words = b->d_words; fa.meta = b->d_meta; fa.worklist = in_list; fa.nwork_dev = in_count; fa.nwork = in_n;
      fa.score = b->d_score; fa.status = b->d_status; fa.fb_list = out_list; fa.fb_count = out_count;
      fa.ef = (b->dcfg.endsfree && (b->dcfg.pbf | b->dcfg.pef | b->dcfg.tbf | b->dcfg.tef)) ? 1 : 0;
      fa.pbf = b->dcfg.pbf; fa.pef = b->dcfg.pef; fa.tbf = b->dcfg.tbf; fa.tef = b->dcfg.tef;
      fa.heur = b->dcfg.heuristic; fa.min_wf_len = b->dcfg.min_wf_len; fa.max_dist_thr = b->dcfg.max_dist_thr;
      fa.steps_between = b->dcfg.steps_between; fa.max_steps = b->dcfg.max_steps; fa.xdrop = b->dcfg.xdrop; fa.scope = b->dcfg.scope;
      if (wfa::launch_seg_heur(b->dcfg, al->cu_count, knob(al, K_FAST_WAVES_PER_CU, 256), stream, fa) != 0) {
        al->err = "segmented kernel launch failed"; return WFA_HIP_EDEVICE;
      }
      if (first_stage) b->last_kernel_pairs = in_n;
      in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
    }
    DualStream pending_walks{al, stream};   // walks of a split stage left running under the band stages behind it
    bool pipe_tail = false;                 // the stages behind the split stage ran beside it, the general kernel included
    std::chrono::steady_clock::time_point band_prev;
    if (knob(al, K_STAGE_TIMING, 0) != 0) { (void)hipStreamSynchronize(stream); band_prev = std::chrono::steady_clock::now(); }
    for (int i = 0; i < n_stages; ++i) {
      wfa::BandArgs ba;
      memset(&ba, 0, sizeof(ba));
      ba.words = b->d_words; ba.meta = b->d_meta; ba.worklist = in_list; ba.nwork_dev = in_count; ba.nwork = in_n;
      ba.score = b->d_score; ba.status = b->d_status;
      ba.cigar_ops = b->d_ops; ba.cigar_off = b->d_cigar_off; ba.cigar_begin = b->d_cigar_begin; ba.cigar_len = b->d_cigar_len;
      uint32_t* out_list = b->d_fb_list2[out_sel];
      uint32_t* out_count = next_count();
      ba.fb_list = out_list; ba.fb_count = out_count;
      ba.g = wfa::band_gcd(b->dcfg, b->ncomp == 5);
      ba.x = b->dcfg.x; ba.oe = b->dcfg.o1 + b->dcfg.e1; ba.e = b->dcfg.e1;
      if (b->ncomp == 5) { ba.oe2 = b->dcfg.o2 + b->dcfg.e2; ba.e2 = b->dcfg.e2; }
      ba.min_wf_len = b->dcfg.min_wf_len; ba.max_dist_thr = b->dcfg.max_dist_thr; ba.steps_between = b->dcfg.steps_between;
      ba.heur = b->dcfg.heuristic; ba.xdrop = b->dcfg.xdrop; ba.max_steps = b->dcfg.max_steps; ba.scope = b->dcfg.scope;
      const int words = ((b->max_len + 15) >> 4) + 4;
      // (round 5: up to 13 KB per wave — reads of up to 26 kb: the 128-diagonal slim kernel then runs four waves per SIMD instead of
      // seven and is still twice as fast as wfa_band_kernel on HBM-resident sequences (15 kb: 3.1 -> 5.9 M aln/s score); beyond, the
      // waves a CU can hold are too few)
      bool seqlds = ((size_t)words * 8 <= (size_t)knob(al, K_BAND_LDS_MAX, 13312)) && knob(al, K_BAND_NO_LDS, 0) == 0;
      ba.lds_words = seqlds ? words : 0;
      // longer reads, wf-adaptive or no heuristic, gap-affine: wfa_slim_kernel on WINDOWS of the sequences (10 k bases of each in LDS,
      // moved along as the alignment advances) — launches it does not take (slim_takes) read the sequences from HBM as before
      const bool try_win = !seqlds && b->ncomp == 3 && b->dcfg.heuristic != WFA_HEUR_XDROP && knob(al, K_BAND_NO_LDS, 0) == 0 && knob(al, K_BAND_NO_WIN, 0) == 0;
      ba.debug = knob(al, K_BAND_DEBUG, 0);
      ba.slim = knob(al, K_BAND_SLIM, 1);   // (0: wfa_band_kernel also where wfa_slim_kernel would take the launch)
      ba.pb_raw = 0;
      if (knob(al, K_STAGE_TIMING, 0) != 0 && b->max_len > WFA_FAST_MAX_LEN) {   // (counting builds; [8..15] belong to the short-read stages otherwise)
        ba.dbg = b->d_counters + 8;
        (void)hipMemsetAsync(b->d_counters + 8, 0, 8 * sizeof(uint32_t), stream);
      }
      ba.h16 = (b->max_len < 32000) ? 1 : 0;
      ba.ef = (b->dcfg.endsfree && (b->dcfg.pbf | b->dcfg.pef | b->dcfg.tbf | b->dcfg.tef)) ? 1 : 0;
      ba.pbf = b->dcfg.pbf; ba.pef = b->dcfg.pef; ba.tbf = b->dcfg.tbf; ba.tef = b->dcfg.tef;
      ba.hist = (i > 0 && later_off) ? al->ws + later_off / 4 : al->ws; ba.hist_stride = band_stride[i];
      const bool split = full && in_count == nullptr && b->max_len > knob(al, K_BAND_SPLIT_MIN, 100) && knob(al, K_BAND_NO_SPLIT, 0) == 0;
      if (split) {
        // history slot per PAIR: as many pairs per launch as the workspace holds; the walks of a launch run
        // afterwards in a thread-per-alignment kernel
        const int64_t slot_ints = (pb_mode && pb_stride) ? pb_stride : band_stride[i];
        if (pb_mode && pb_stride) { ba.pb = 1; ba.hist_stride = pb_stride; ba.pb_code_ints = pb_code_ints; ba.pb_event_ints = pb_event_ints; }
        const int64_t slot_bytes = slot_ints * 4 + (int64_t)sizeof(int4);
        // Slots: `cap` pairs fit the stage's region.  The launches are balanced (no small last launch) and their walks run on
        // the side stream under the next alignment kernel: every launch has slots of its own when the whole batch fits
        // (two launches at least, if each still fills the chip a few times over), otherwise the launches alternate between
        // two halves of the region (only while a half still fills the chip: gap-affine-2p with explicit history at 10 kb
        // runs 2 waves per SIMD and its launches are small already — halving them measured -35 %).
        const size_t region = later_off ? later_off : al->ws_bytes;
        const int64_t cap = std::min<int64_t>((int64_t)(region / (size_t)slot_bytes), in_n);
        if (cap < 1) { al->err = "band history does not fit"; return WFA_HIP_EDEVICE; }
        const int64_t chip = (int64_t)al->cu_count * 32;
        const bool dual_ok = knob(al, K_NO_DUAL, 0) == 0;
        const bool all_fit = cap >= (int64_t)in_n;
        int64_t nl, per_launch;
        bool halves = false;
        if (all_fit) {
          nl = (dual_ok && (int64_t)in_n >= 4 * chip) ? 2 : 1;
          per_launch = ((int64_t)in_n + nl - 1) / nl;
        } else {
          halves = dual_ok && cap / 2 >= chip;
          const int64_t lmax = halves ? cap / 2 : cap;
          nl = ((int64_t)in_n + lmax - 1) / lmax;
          per_launch = std::min<int64_t>(lmax, (((int64_t)in_n + nl - 1) / nl + 63) & ~63ll);
        }
        // the pipelined tail (see want_pipe_tail): this stage's launches and, beside each, the later stages on its leftovers
        pipe_tail = want_pipe_tail && i == 0 && gen_off != 0 && nl >= 2 && nl <= 90 && !tile_stage.on && n_wide == 0 && (all_fit || halves);
        uint32_t* const snap_a = b->d_counters + 64;    // [j] = this stage's hand-over count before launch j
        uint32_t* const snap_b = b->d_counters + 160;   // the same for the stage behind it
        uint32_t* tail_list = nullptr; uint32_t* tail_count = nullptr;
        wfa::BandArgs tb = ba;   // the stage behind: everything but lists, history region and window width is this stage's
        if (pipe_tail) {
          if (!al->tail_join) {
            for (int q = 0; q < 2; ++q) HIP_TRY(al, hipEventCreateWithFlags(&al->tail_fork[q], hipEventDisableTiming));
            HIP_TRY(al, hipEventCreateWithFlags(&al->tail_join, hipEventDisableTiming));
          }
          tail_list = b->d_fb_list2[out_sel ^ 1]; tail_count = next_count();
          tb.worklist = out_list; tb.fb_list = tail_list; tb.fb_count = tail_count;
          tb.hist = al->ws + later_off / 4; tb.hist_stride = band_stride[1];
          tb.pb = 0; tb.pb_code_ints = 0; tb.pb_event_ints = 0; tb.pb_raw = 0; tb.split = 0;   // (unsplit: explicit history, walked in-kernel)
          // (the tail stream's first kernel must see the counters zeroed and the previous run's use of the regions over)
          HIP_TRY(al, hipEventRecord(al->tail_fork[0], stream));
          HIP_TRY(al, hipStreamWaitEvent(al->tail_stream, al->tail_fork[0], 0));
        }
        ba.split = 1;
        char* const es_base = reinterpret_cast<char*>(al->ws) + (size_t)cap * slot_ints * 4;   // end states behind the slots
        DualStream dual{al, stream};
        // (one launch: its walks go to the side stream only if a band stage with a region of its own follows)
        const bool side = (nl > 1 && (all_fit || halves)) || (nl == 1 && later_off != 0 && i + 1 < n_stages);
        { const int drc = dual.begin(side); if (drc != WFA_HIP_OK) return drc; }
        int64_t launch = 0;
        for (int64_t w0 = 0; w0 < in_n; w0 += per_launch, ++launch) {
          const uint32_t cnt = (uint32_t)std::min<int64_t>(per_launch, in_n - w0);
          const int64_t slot0 = all_fit ? w0 : ((halves && (launch & 1)) ? per_launch : 0);
          ba.hist = al->ws + slot0 * slot_ints;
          ba.end_state = reinterpret_cast<int4*>(es_base) + slot0;
          ba.work_begin = (uint32_t)w0; ba.nwork = cnt;
          const long long grid = std::min<long long>((long long)al->cu_count * knob(al, K_BAND_WAVES_PER_CU, 128), cnt);
          if (!all_fit) { const int drc = dual.before_align(launch); if (drc != WFA_HIP_OK) return drc; }
          if (try_win && !ba.win) {   // (decided once per stage: the launches of a stage share everything slim_takes looks at)
            wfa::BandArgs t = ba; t.win = 1; t.lds_words = 643;
            if (wfa::slim_launches(t, band_nch[i], full, adapt, true)) { ba.win = 1; ba.lds_words = 643; seqlds = true; }
          }
          ba.pb_raw = wfa::slim_launches(ba, band_nch[i], full, adapt, seqlds) ? 1 : 0;   // (wfa_slim_kernel writes comparison bits; the walk below decodes them)
          if (wfa::launch_band(ba, band_nch[i], full, adapt, seqlds, grid, stream) != 0) { al->err = "band kernel launch failed"; return WFA_HIP_EDEVICE; }
          if (pipe_tail) {
            // [snap_a[launch], snap_a[launch + 1]) = what this launch handed on (snap_a[0] = 0: the counters are zeroed per run)
            hipLaunchKernelGGL(wfa_snapshot_kernel, dim3(1), dim3(1), 0, stream, out_count, snap_a + launch + 1);
            HIP_TRY(al, hipEventRecord(al->tail_fork[launch & 1], stream));
            HIP_TRY(al, hipStreamWaitEvent(al->tail_stream, al->tail_fork[launch & 1], 0));
            tb.wbeg_dev = snap_a + launch; tb.nwork_dev = snap_a + launch + 1; tb.nwork = in_n;
            if (wfa::launch_band(tb, band_nch[1], full, adapt, seqlds, band_grid[1], al->tail_stream) != 0) { al->err = "band kernel launch failed"; return WFA_HIP_EDEVICE; }
            hipLaunchKernelGGL(wfa_snapshot_kernel, dim3(1), dim3(1), 0, al->tail_stream, tail_count, snap_b + launch + 1);
            const int grc = launch_general_dyn(al, b, al->tail_stream, true, tail_list, snap_b + launch + 1, in_n, gen_stride, gen_grid, g.threads,
                                               b->d_ovf_list[0], b->d_counters + 1, reinterpret_cast<int32_t*>(reinterpret_cast<char*>(al->ws) + gen_off), snap_b + launch);
            if (grc != WFA_HIP_OK) return grc;
          }
          hipStream_t ws_ = stream;
          { const int drc = dual.walk_stream(launch, &ws_); if (drc != WFA_HIP_OK) return drc; }
          if (wfa::launch_band_bt(ba, band_nch[i], ws_) != 0) { al->err = "band backtrace launch failed"; return WFA_HIP_EDEVICE; }
          { const int drc = dual.after_walk(launch); if (drc != WFA_HIP_OK) return drc; }
        }
        // the walks still running are joined before anything reuses their slots: at once, unless the next band stage has
        // its own region
        if (later_off && i + 1 < n_stages && dual.on) pending_walks = dual;
        else { const int drc = dual.end(); if (drc != WFA_HIP_OK) return drc; }
        if (pipe_tail) {
          // everything behind this stage has been enqueued beside it: the main stream joins the tail stream, the stage behind is
          // skipped below and so is the last launch of the general kernel; the lists move on as if the stages had run here
          HIP_TRY(al, hipEventRecord(al->tail_join, al->tail_stream));
          HIP_TRY(al, hipStreamWaitEvent(stream, al->tail_join, 0));
          if (first_stage) b->last_kernel_pairs = in_n;
          in_list = tail_list; in_count = tail_count; first_stage = false;   // (out_sel: two lists were used, the selector is where it was)
          break;
        }
      } else {
        if (try_win) {
          wfa::BandArgs t = ba; t.win = 1; t.lds_words = 643;
          if (wfa::slim_launches(t, band_nch[i], full, adapt, true)) { ba.win = 1; ba.lds_words = 643; seqlds = true; }
        }
        if (wfa::launch_band(ba, band_nch[i], full, adapt, seqlds, band_grid[i], stream) != 0) { al->err = "band kernel launch failed"; return WFA_HIP_EDEVICE; }
      }
      if (knob(al, K_STAGE_TIMING, 0) != 0) {  // development aid: synchronises (the time of the stage is the gap between these lines' events)
        (void)hipStreamSynchronize(stream);
        uint32_t handed = 0, took = in_n; (void)hipMemcpy(&handed, out_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (in_count) (void)hipMemcpy(&took, in_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
        const auto t_now = std::chrono::steady_clock::now();
        fprintf(stderr, "[wfa_hip] band stage %d (%d diagonals): took %u pairs, handed on %u; %.3f ms\n", i, 64 * band_nch[i], took, handed,
                std::chrono::duration<double, std::milli>(t_now - band_prev).count());
        if (ba.dbg) {
          uint32_t dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpy(dbg, ba.dbg, sizeof(dbg), hipMemcpyDeviceToHost);
          if (dbg[0] | dbg[1]) fprintf(stderr, "[wfa_hip]   slim counters: steps of 64 diagonals %u, of 128 %u, extension rounds %u, cut-offs that cut %u, window shifts %u, end trims %u, live diagonals / 64 %u\n",
                                       dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6]);
        }
        band_prev = std::chrono::steady_clock::now();
      }
      if (first_stage) b->last_kernel_pairs = in_n;
      in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
    }
    { const int drc = pending_walks.end(); if (drc != WFA_HIP_OK) return drc; }   // (the stages below use the workspace from its start)
    if (tile_stage.on) {
      wfa::TileArgs& ta = tile_stage.a;
      uint32_t* out_list = b->d_fb_list2[out_sel];
      uint32_t* out_count = next_count();
      ta.words = b->d_words; ta.meta = b->d_meta; ta.worklist = in_list; ta.nwork_dev = in_count; ta.nwork = in_n;
      ta.score = b->d_score; ta.status = b->d_status; ta.fb_list = out_list; ta.fb_count = out_count;
      ta.cigar_ops = b->d_ops; ta.cigar_off = b->d_cigar_off; ta.cigar_begin = b->d_cigar_begin; ta.cigar_len = b->d_cigar_len;
      ta.rows = reinterpret_cast<short*>(al->ws);
      ta.hist = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(al->ws) + tile_stage.hist_off);
      ta.ef = b->dcfg.endsfree ? 1 : 0;
      ta.pbf = b->dcfg.pbf; ta.pef = b->dcfg.pef; ta.tbf = b->dcfg.tbf; ta.tef = b->dcfg.tef;
      ta.max_steps = b->dcfg.max_steps;
      hipEvent_t se0 = nullptr, se1 = nullptr;
      const bool stage_timing = knob(al, K_STAGE_TIMING, 0) != 0;
      if (stage_timing) { (void)hipEventCreate(&se0); (void)hipEventCreate(&se1); (void)hipEventRecord(se0, stream); }
      ta.dbg = stage_timing ? b->d_counters + 8 : nullptr;   // ([8] is the lane-full stage's list count: short reads only)
      if (stage_timing) (void)hipMemsetAsync(b->d_counters + 8, 0, 8 * sizeof(uint32_t), stream);
      if (wfa::launch_tile(full, wide_two, ta, tile_stage.grid, tile_stage.threads, tile_stage.smem, stream, tile_stage.w32) != 0) { al->err = "tile kernel launch failed"; return WFA_HIP_EDEVICE; }
      if (stage_timing) {  // development aid: synchronises
        (void)hipEventRecord(se1, stream); (void)hipEventSynchronize(se1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, se0, se1);
        uint32_t handed = 0; (void)hipMemcpy(&handed, out_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
        uint32_t dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpy(dbg, b->d_counters + 8, sizeof(dbg), hipMemcpyDeviceToHost);
        if (dbg[3] | dbg[4]) fprintf(stderr, "[wfa_hip] tile profile (k-cycles of s_memtime, summed over waves): load %u, steps %u, write-back %u; steps that entered the end test %u, the long-match loop %u\n", dbg[3], dbg[4], dbg[5], dbg[6], dbg[7]);
        fprintf(stderr, "[wfa_hip] tile stage (T %d, Wt %d, %d x %d threads, %zu B LDS): %.3f ms, handed on %u pairs; %u super-steps, %u repeated with statistics, %u tiles\n",
                ta.g.T, ta.g.Wt, tile_stage.grid, tile_stage.threads, tile_stage.smem, ms, handed, dbg[0], dbg[1], dbg[2]);
        (void)hipEventDestroy(se0); (void)hipEventDestroy(se1);
      }
      if (first_stage) b->last_kernel_pairs = in_n;
      in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
    }
    for (int ws_i = 0; ws_i < n_wide; ++ws_i) {
      WideStage& st = wide_stage[ws_i];
      wfa::WideArgs& wa = st.a;
      uint32_t* out_list = b->d_fb_list2[out_sel];
      uint32_t* out_count = next_count();
      wa.words = b->d_words; wa.meta = b->d_meta; wa.worklist = in_list; wa.nwork_dev = in_count; wa.nwork = in_n;
      wa.score = b->d_score; wa.status = b->d_status; wa.fb_list = out_list; wa.fb_count = out_count;
      wa.cigar_ops = b->d_ops; wa.cigar_off = b->d_cigar_off; wa.cigar_begin = b->d_cigar_begin; wa.cigar_len = b->d_cigar_len;
      wa.rows = st.grows ? reinterpret_cast<short*>(al->ws) : nullptr;
      wa.hist = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(al->ws) + st.hist_off);
      wa.ef = b->dcfg.endsfree ? 1 : 0;
      wa.pbf = b->dcfg.pbf; wa.pef = b->dcfg.pef; wa.tbf = b->dcfg.tbf; wa.tef = b->dcfg.tef;
      wa.max_steps = b->dcfg.max_steps;
      wa.heur = (b->dcfg.heuristic == WFA_HEUR_ADAPTIVE) ? 1 : 0; wa.min_wf_len = b->dcfg.min_wf_len; wa.max_dist_thr = b->dcfg.max_dist_thr;
      wa.steps_between = b->dcfg.steps_between;
      hipEvent_t se0 = nullptr, se1 = nullptr;
      const bool stage_timing = knob(al, K_STAGE_TIMING, 0) != 0;
      if (stage_timing) { (void)hipEventCreate(&se0); (void)hipEventCreate(&se1); (void)hipEventRecord(se0, stream); }
      if (wfa::launch_wide(full, wide_two, st.grows, wa, st.grid, st.threads, st.smem, stream, st.w32) != 0) { al->err = "wide kernel launch failed"; return WFA_HIP_EDEVICE; }
      if (stage_timing) {  // development aid: synchronises
        (void)hipEventRecord(se1, stream); (void)hipEventSynchronize(se1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, se0, se1);
        uint32_t handed = 0; (void)hipMemcpy(&handed, out_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
        fprintf(stderr, "[wfa_hip] wide stage %d (%d x %d threads): %.3f ms, handed on %u pairs\n", ws_i, st.grid, st.threads, ms, handed);
        (void)hipEventDestroy(se0); (void)hipEventDestroy(se1);
      }
      if (first_stage) b->last_kernel_pairs = in_n;
      in_list = out_list; in_count = out_count; out_sel ^= 1; first_stage = false;
    }
    b->leftover_count = in_count;
    // (lin: what the register stages finished carries raw scores, what the general kernel finishes below — under the original
    // configuration — final ones: the translation comes here, before it; a pair still to be aligned carries a non-zero status)
    if (lin && b->dcfg.score_mode != 0)
      hipLaunchKernelGGL(wfa_score_translate_kernel, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, stream, b->d_score, b->d_status, b->d_meta,
                         (long long)b->n, b->dcfg.score_mode, b->dcfg.sw_match);
    if (!pipe_tail) {   // (the pipelined tail has run the general kernel on every launch's leftovers already)
      rc = launch_general_dyn(al, b, stream, true, in_list, in_count, in_n, g.ws_stride, g.grid, g.threads,
                              b->d_ovf_list[0], b->d_counters + 1);
      if (rc != WFA_HIP_OK) return rc;
    }
    if (first_stage) b->last_kernel_pairs = in_n;
    { const int drc = lane_expands.end(); if (drc != WFA_HIP_OK) return drc; }
    if (use_lanefull) {   // what the walks of the lane-full stage handed on (nothing, as a rule)
      rc = launch_general_dyn(al, b, stream, true, reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(al->ws) + lanefull_list_off), b->d_counters + 8,
                              (uint32_t)b->n_packed, g.ws_stride, g.grid, g.threads, b->d_ovf_list[0], b->d_counters + 1);
      if (rc != WFA_HIP_OK) return rc;
    }
  }
  HIP_TRY(al, hipEventRecord(ev1, stream));
  // 3) 8-bit pairs (non-ACGT letters, wildcard matching)
  if (b->n_bytes > 0) {
    Geometry g = plan_general(al, b, b->n_bytes, b->arena_fixed + b->arena_ints);
    int rc = ensure_ws(al, (size_t)g.grid * g.ws_stride * 4);
    if (rc != WFA_HIP_OK) return rc;
    rc = launch_general_dyn(al, b, stream, false, b->d_list_bytes, nullptr, b->n_bytes, g.ws_stride, g.grid, g.threads,
                            b->d_ovf_list[0], b->d_counters + 1);
    if (rc != WFA_HIP_OK) return rc;
  }
  // scores of a configuration that was mapped onto another (match < 0, the one-component distances): translated in-stream, in every
  // scope (round 6: full scope used to wait for wfa_hip_batch_sync — an extra launch and a second synchronisation; now only the pairs an
  // arena overflow re-runs are translated there: a pair that is still to be re-run carries a non-zero status and is skipped here)
  if (b->dcfg.score_mode != 0 && !b->dcfg.lin)
    hipLaunchKernelGGL(wfa_score_translate_kernel, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, stream, b->d_score, b->d_status, b->d_meta,
                       (long long)b->n, b->dcfg.score_mode, b->dcfg.sw_match);
  HIP_TRY(al, hipEventRecord(al->ws_event, stream));
  al->ws_event_recorded = true; al->ws_last_stream = stream;
  side_guard.ok = true;
  return WFA_HIP_OK;
}

// Re-run the pairs whose arena overflowed with an 8x larger arena (fewer workgroups in flight), until
// none is left or the arena no longer fits the device: those get WF_STATUS_OOM (-200, wfa.h:50).
static int retry_overflows(wfa_hip_batch* b) {
  wfa_hip_aligner* al = b->al;
  hipStream_t stream = b->last_stream;
  int cur = 0;
  for (int round = 0; round < 12; ++round) {
    uint32_t novf = 0;
    HIP_TRY(al, hipMemcpy(&novf, b->d_counters + 1 + cur, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (novf == 0) return WFA_HIP_OK;
    std::vector<uint32_t> ids(novf);
    HIP_TRY(al, hipMemcpy(ids.data(), b->d_ovf_list[cur], novf * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint8_t> flags((size_t)b->n, 0);
    if (b->n_bytes) HIP_TRY(al, hipMemcpy(flags.data(), b->d_flags, (size_t)b->n, hipMemcpyDeviceToHost));
    const bool all_bytes = (b->cfg.wildcard >= 0 && b->wild < 0);
    std::vector<uint32_t> lp, lb;
    for (uint32_t id : ids) ((all_bytes || flags[id]) ? lb : lp).push_back(id);
    b->arena_ints *= 8;
    const int nxt = cur ^ 1;
    HIP_TRY(al, hipMemset(b->d_counters + 1 + nxt, 0, sizeof(uint32_t)));
    const int64_t budget = free_budget(al);
    if ((b->arena_fixed + b->arena_ints) * 4 > budget) {
      // cannot grow further: report OOM for these pairs
      std::vector<int32_t> st(1, WFA_STATUS_OOM), sc(1, INT_MIN);
      for (uint32_t id : ids) {
        HIP_TRY(al, hipMemcpy(b->d_status + id, st.data(), 4, hipMemcpyHostToDevice));
        HIP_TRY(al, hipMemcpy(b->d_score + id, sc.data(), 4, hipMemcpyHostToDevice));
        if (b->d_cigar_len) HIP_TRY(al, hipMemset(b->d_cigar_len + id, 0, 4));
      }
      return WFA_HIP_OK;
    }
    for (int kind = 0; kind < 2; ++kind) {
      std::vector<uint32_t>& l = kind ? lb : lp;
      if (l.empty()) continue;
      uint32_t* d_l = b->d_fb_list2[0];  // free to reuse: the first pass is complete
      HIP_TRY(al, hipMemcpy(d_l, l.data(), l.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      Geometry g = plan_general(al, b, (uint32_t)l.size(), b->arena_fixed + b->arena_ints);
      int rc = ensure_ws(al, (size_t)g.grid * g.ws_stride * 4);
      if (rc != WFA_HIP_OK) return rc;
      rc = launch_general_dyn(al, b, stream, kind == 0, d_l, nullptr, (uint32_t)l.size(), g.ws_stride, g.grid, g.threads,
                              b->d_ovf_list[nxt], b->d_counters + 1 + nxt);
      if (rc != WFA_HIP_OK) return rc;
      if (b->dcfg.score_mode != 0 && !b->dcfg.lin)   // (the stream's translation pass skipped these pairs: their status was not 0 then; lin: the general kernel's scores are final)
        hipLaunchKernelGGL(wfa_score_translate_list_kernel, dim3((unsigned)((l.size() + 255) / 256)), dim3(256), 0, stream, d_l, (uint32_t)l.size(),
                           b->d_score, b->d_status, b->d_meta, b->dcfg.score_mode, b->dcfg.sw_match);
      HIP_TRY(al, hipStreamSynchronize(stream));
    }
    cur = nxt;
  }
  al->err = "arena overflow retries exhausted";
  return WFA_HIP_EDEVICE;
}

extern "C" int wfa_hip_batch_sync(wfa_hip_batch_t* b) {
  if (!b) return WFA_HIP_EINVAL;
  wfa_hip_aligner* al = b->al;
  if (!b->ran || b->synced) return WFA_HIP_OK;
  HIP_TRY(al, hipSetDevice(al->device));
  HIP_TRY(al, hipStreamSynchronize(b->last_stream));
  if (b->n > 0) {
    b->ms_sum = 0.0; b->ms_runs = 0;
    for (size_t i = 0; i + 1 < b->ev_used; i += 2) {
      float ms = 0.f;
      HIP_TRY(al, hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]));
      b->ms_sum += ms; b->ms_runs += 1;
    }
    b->last_ms = b->ms_runs ? (float)(b->ms_sum / b->ms_runs) : 0.f;
    b->ev_used = 0; b->runs_pending = 0;
    uint32_t fb = 0;
    if (b->leftover_count) HIP_TRY(al, hipMemcpy(&fb, b->leftover_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
    b->last_fallback = fb;
    if (b->cfg.scope == WFA_SCOPE_FULL) {
      const int rc = retry_overflows(b);
      if (rc != WFA_HIP_OK) return rc;
    }
  }
  b->synced = true;
  return WFA_HIP_OK;
}

extern "C" int wfa_hip_batch_results(wfa_hip_batch_t* b, int32_t* score, int32_t* status, uint8_t* cigar_ops,
                                     const int64_t* cigar_off, int64_t* cigar_begin, int32_t* cigar_len) {
  if (!b) return WFA_HIP_EINVAL;
  wfa_hip_aligner* al = b->al;
  int rc = wfa_hip_batch_sync(b);
  if (rc != WFA_HIP_OK) return rc;
  const int64_t n = b->n;
  if (n == 0) return WFA_HIP_OK;
  if (!score || !status) { al->err = "score/status outputs are required"; return WFA_HIP_EINVAL; }
  HIP_TRY(al, hipMemcpy(score, b->d_score, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(al, hipMemcpy(status, b->d_status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  const bool full = (b->cfg.scope == WFA_SCOPE_FULL);
  if (cigar_len) {
    if (full) HIP_TRY(al, hipMemcpy(cigar_len, b->d_cigar_len, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    else memset(cigar_len, 0, (size_t)n * sizeof(int32_t));
  }
  if (cigar_begin) {
    if (full) HIP_TRY(al, hipMemcpy(cigar_begin, b->d_cigar_begin, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    else memset(cigar_begin, 0, (size_t)n * sizeof(int64_t));
  }
  if (full && cigar_ops) {
    if (!cigar_off || !cigar_begin || !cigar_len) { al->err = "cigar_off/cigar_begin/cigar_len are required with cigar_ops"; return WFA_HIP_EINVAL; }
    // the device regions are laid out back to back in pair order (plen+tlen bytes each); the caller's
    // regions may be spaced differently: copy whole, then re-base per pair when they differ
    // (a shard of a larger batch: the caller's regions are the device layout shifted by cigar_off[0])
    bool same = true;
    int64_t acc = 0;
    const int64_t base0 = cigar_off[0];
    for (int64_t i = 0; i < n && same; ++i) { same = (cigar_off[i] == base0 + acc); acc += b->h_plen[i] + b->h_tlen[i]; }
    if (same) {
      if (b->ops_bytes >= ((int64_t)64 << 20) && knob(al, K_NO_PIPE, 0) == 0) {   // (long reads: through the pinned ring, the host copies on a team of threads)
        const int drc = staged_download(al, cigar_ops + base0, b->d_ops, (size_t)b->ops_bytes, b->last_stream ? b->last_stream : al->stream);
        if (drc != WFA_HIP_OK) return drc;
      } else if (b->ops_bytes) HIP_TRY(al, hipMemcpy(cigar_ops + base0, b->d_ops, (size_t)b->ops_bytes, hipMemcpyDeviceToHost));
      if (base0) for (int64_t i = 0; i < n; ++i) cigar_begin[i] += base0;
    } else {
      std::vector<uint8_t> tmp((size_t)std::max<int64_t>(b->ops_bytes, 1));
      if (b->ops_bytes) HIP_TRY(al, hipMemcpy(tmp.data(), b->d_ops, (size_t)b->ops_bytes, hipMemcpyDeviceToHost));
      acc = 0;
      for (int64_t i = 0; i < n; ++i) {
        const int64_t cap = (int64_t)b->h_plen[i] + b->h_tlen[i];
        const int64_t rel = cigar_begin[i] - acc;
        if (cigar_len[i] > 0) memcpy(cigar_ops + cigar_off[i] + rel, tmp.data() + cigar_begin[i], (size_t)cigar_len[i]);
        cigar_begin[i] = cigar_off[i] + rel;
        acc += cap;
      }
    }
  }
  return WFA_HIP_OK;
}

extern "C" int wfa_hip_batch_last_kernel_ms(wfa_hip_batch_t* b, float* ms, int64_t* pairs) {
  if (!b) return WFA_HIP_EINVAL;
  const int rc = wfa_hip_batch_sync(b);
  if (rc != WFA_HIP_OK) return rc;
  if (ms) *ms = b->last_ms;
  if (pairs) *pairs = b->last_kernel_pairs;
  return WFA_HIP_OK;
}

extern "C" int64_t wfa_hip_batch_algorithmic_bytes(const wfa_hip_batch_t* b) {
  if (!b) return 0;
  int64_t bytes = b->packed_bytes + 8 * b->n;
  if (b->cfg.scope == WFA_SCOPE_FULL) bytes += b->ops_bytes;
  return bytes;
}

// ---- device-side cigartuples + locations ----------------------------------------------------------
extern "C" int64_t wfa_hip_batch_rle_counts(wfa_hip_batch_t* b, int32_t* run_count, int32_t* locations) {
  if (!b) return WFA_HIP_EINVAL;
  wfa_hip_aligner* al = b->al;
  if (b->cfg.scope != WFA_SCOPE_FULL) { al->err = "run-length encoding needs scope=full"; return WFA_HIP_EINVAL; }
  int rc = wfa_hip_batch_sync(b);
  if (rc != WFA_HIP_OK) return rc;
  const int64_t n = b->n;
  b->rle_total = 0;
  if (n == 0) return 0;
  const size_t nn = (size_t)n;
  if (!b->d_plen) {
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_plen, nn * 4)); HIP_TRY(al, pool_alloc(al, (void**)&b->d_tlen, nn * 4));
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_run_count, nn * 4)); HIP_TRY(al, pool_alloc(al, (void**)&b->d_locs, nn * 16));
    HIP_TRY(al, pool_alloc(al, (void**)&b->d_run_off, (nn + 1) * 8));
    HIP_TRY(al, hipMemcpy(b->d_plen, b->h_plen.data(), nn * 4, hipMemcpyHostToDevice));
    HIP_TRY(al, hipMemcpy(b->d_tlen, b->h_tlen.data(), nn * 4, hipMemcpyHostToDevice));
  }
  const int grid = (int)std::min<int64_t>((n + 3) / 4, (int64_t)al->cu_count * 16);
  hipLaunchKernelGGL(wfa::wfa_rle_kernel, dim3(grid), dim3(256), 0, al->stream, b->d_ops, b->d_cigar_begin, b->d_cigar_len,
                     b->d_plen, b->d_tlen, n, b->d_run_count, b->d_locs, (const int64_t*)nullptr, (uint8_t*)nullptr, (int32_t*)nullptr);
  HIP_TRY(al, hipGetLastError());
  std::vector<int32_t> cnt(nn);
  HIP_TRY(al, hipMemcpyAsync(cnt.data(), b->d_run_count, nn * 4, hipMemcpyDeviceToHost, al->stream));
  if (locations) HIP_TRY(al, hipMemcpyAsync(locations, b->d_locs, nn * 16, hipMemcpyDeviceToHost, al->stream));
  HIP_TRY(al, hipStreamSynchronize(al->stream));
  std::vector<int64_t> off(nn + 1);
  off[0] = 0;
  for (size_t i = 0; i < nn; ++i) off[i + 1] = off[i] + cnt[i];
  HIP_TRY(al, hipMemcpy(b->d_run_off, off.data(), (nn + 1) * 8, hipMemcpyHostToDevice));
  if (run_count) memcpy(run_count, cnt.data(), nn * 4);
  b->rle_total = off[nn];
  return b->rle_total;
}

extern "C" int wfa_hip_batch_rle_runs(wfa_hip_batch_t* b, uint8_t* run_code, int32_t* run_len) {
  if (!b) return WFA_HIP_EINVAL;
  wfa_hip_aligner* al = b->al;
  if (b->rle_total < 0) { al->err = "call wfa_hip_batch_rle_counts first"; return WFA_HIP_EINVAL; }
  const int64_t n = b->n, total = b->rle_total;
  if (n == 0 || total == 0) return WFA_HIP_OK;
  if (!run_code || !run_len) { al->err = "null output"; return WFA_HIP_EINVAL; }
  uint8_t* d_code = nullptr; int32_t* d_start = nullptr;
  HIP_TRY(al, pool_alloc(al, (void**)&d_code, (size_t)total));
  HIP_TRY(al, pool_alloc(al, (void**)&d_start, (size_t)total * 4));
  const int grid = (int)std::min<int64_t>((n + 3) / 4, (int64_t)al->cu_count * 16);
  hipLaunchKernelGGL(wfa::wfa_rle_kernel, dim3(grid), dim3(256), 0, al->stream, b->d_ops, b->d_cigar_begin, b->d_cigar_len,
                     b->d_plen, b->d_tlen, n, b->d_run_count, b->d_locs, (const int64_t*)b->d_run_off, d_code, d_start);
  HIP_TRY(al, hipGetLastError());
  HIP_TRY(al, hipMemcpyAsync(run_code, d_code, (size_t)total, hipMemcpyDeviceToHost, al->stream));
  HIP_TRY(al, hipMemcpyAsync(run_len, d_start, (size_t)total * 4, hipMemcpyDeviceToHost, al->stream));  // starts for now
  std::vector<int32_t> cnt((size_t)n), clen((size_t)n);
  HIP_TRY(al, hipMemcpyAsync(cnt.data(), b->d_run_count, (size_t)n * 4, hipMemcpyDeviceToHost, al->stream));
  HIP_TRY(al, hipMemcpyAsync(clen.data(), b->d_cigar_len, (size_t)n * 4, hipMemcpyDeviceToHost, al->stream));
  HIP_TRY(al, hipStreamSynchronize(al->stream));
  pool_release(al, d_code); pool_release(al, d_start);
  // run length = next run's start (or the end of the op string) - this run's start
  int64_t r = 0;
  for (int64_t i = 0; i < n; ++i) {
    for (int32_t j = 0; j < cnt[i]; ++j, ++r) {
      const int32_t next = (j + 1 < cnt[i]) ? run_len[r + 1] : clen[i];
      run_len[r] = next - run_len[r];
    }
  }
  return WFA_HIP_OK;
}

extern "C" int64_t wfa_hip_batch_fallback_pairs(const wfa_hip_batch_t* b) { return b ? b->last_fallback : 0; }

// ---- the single-call path --------------------------------------------------------
// pywfa's usual loop calls the aligner with ONE pair at a time (align.pyx:421-443).  For a handful of pairs the batch
// machinery above (a dozen device arrays, uploads, the pack kernel, several result copies) is most of the call, so such
// calls take this path: everything the kernels read is written by the host into one pinned block and moved to its
// device copy by one small kernel; the general kernel (raw bytes: any alphabet, every configuration) aligns the pairs,
// writing scores / statuses / op bytes straight into the pinned block; one stream synchronisation; the host copies
// the results out.  No allocation, no memcpy call, no per-call environment lookup.
static const int64_t TINY_MAX_PAIRS = 16;          // general-kernel form (a workspace slice per pair)
// banded form: up to 4 096 pairs, whatever the pinned block holds (4 096 x 150 bp with op strings).  Round 3 (was 1 024): the host packs the
// pairs on one thread (~70 ns per pair), so the path scales linearly and meets the batch machinery's fixed ~0.5 ms at ~8 k pairs:
// 2 048 pairs 160 us (score) / 320 us (op strings) against 482 / 730 us, 4 096 pairs 319 / 589 against 534 / 1 001 us
static const int64_t TINY_MAX_PAIRS_BAND = 4096;
static const size_t TINY_IN_BYTES = (size_t)1 << 20, TINY_BLOCK_BYTES = (size_t)4 << 20;

__global__ void __launch_bounds__(256) wfa_tiny_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

// returns 1 when the call was served, 0 when it does not fit this path (the caller takes the batch path), < 0 on error
// (score_mode: the kernels leave -s, as wfa_score_translate_kernel)
static inline int32_t tiny_score(const WfaDevConfig& d, int32_t raw, int32_t status, int plen, int tlen) {
  if (d.score_mode == 0 || status != 0) return raw;
  if (d.score_mode == 2) return -raw;
  return (int32_t)(((long long)d.sw_match * ((long long)plen + tlen) + raw) / 2);
}

// ---- the resident one-pair kernel (wfa_slim.hpp: wfa_slim_kernel_mailbox) ----------------------------------------------
// tell the running instance, if any, to leave, and wait until it has (its last store is alive = 0)
static void mailbox_quit(wfa_hip_aligner* al) {
  if (!al->mb_h) return;
  if (getenv("WFA_HIP_MAILBOX_DEBUG")) fprintf(stderr, "[wfa_hip] mailbox: instance served %u requests, the last one in %.2f us on the device (request %u)\n",
                                               al->mb_h->served, al->mb_h->ticks * 0.01, al->mb_seq);
  if (__atomic_load_n(&al->mb_h->alive, __ATOMIC_ACQUIRE) != 0) {
    __atomic_store_n(&al->mb_h->quit, 1u, __ATOMIC_RELEASE);
    const double t0 = now_ms();
    while (__atomic_load_n(&al->mb_h->alive, __ATOMIC_ACQUIRE) != 0 && now_ms() - t0 < 100.0) __builtin_ia32_pause();
  }
  if (al->mb_stream) (void)hipStreamSynchronize(al->mb_stream);   // (the instance has left, or never started: the stream drains at once)
  __atomic_store_n(&al->mb_h->alive, 0u, __ATOMIC_RELEASE);
  __atomic_store_n(&al->mb_h->quit, 0u, __ATOMIC_RELEASE);
  al->mb_args_valid = false;
}

// one pair through the mailbox: 1 = served (score and raw status in *score_raw / *status_raw, op bytes in the pinned block), 0 = not this
// way (the caller launches a kernel as before)
static int mailbox_call(wfa_hip_aligner* al, const wfa::BandArgs& ba, bool full, bool adapt, const uint32_t* one, size_t one_words,
                        int32_t* score_raw, int32_t* status_raw) {
  if (knob(al, K_MAILBOX, 1) == 0 || al->mb_failures >= 3) return 0;
  if (!al->mb_h) {
    if (hipHostMalloc((void**)&al->mb_h, sizeof(wfa::SlimMailbox), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); al->mb_h = nullptr; al->mb_failures = 3; return 0; }
    memset(al->mb_h, 0, sizeof(wfa::SlimMailbox));
    if (hipHostGetDevicePointer((void**)&al->mb_d, al->mb_h, 0) != hipSuccess ||
        hipStreamCreateWithFlags(&al->mb_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); al->mb_failures = 3; return 0; }
  }
  wfa::SlimMailbox* const mb = al->mb_h;
  wfa::BandArgs want = ba;
  want.one = nullptr; want.done = nullptr;
  // an instance started for another configuration / workspace leaves first
  if (__atomic_load_n(&mb->alive, __ATOMIC_ACQUIRE) != 0 && (!al->mb_args_valid || memcmp(&want, &al->mb_args, sizeof(want)) != 0)) mailbox_quit(al);
  // the request: 15 words of the block + the request number per 64-byte line (the number is the line's own "ready" flag; lines the block
  // does not reach carry the number alone)
  al->mb_seq = (al->mb_seq + 1u) & 0xFFFFFFu;
  const uint32_t k = al->mb_seq;
  for (int ln = 0; ln < WFA_MB_LINES; ++ln) {
    uint32_t line[16];
    for (int j = 0; j < 15; ++j) { const size_t src = (size_t)ln * 15 + j; line[j] = src < one_words ? one[src] : 0u; }
    line[15] = k;
    memcpy(mb->req[ln], line, sizeof(line));
  }
  std::atomic_thread_fence(std::memory_order_release);
  auto answered = [&]() -> bool {
    const unsigned long long d = __atomic_load_n(&mb->done, __ATOMIC_ACQUIRE);
    if ((uint32_t)(d & 0xFFFFFFull) != k) return false;
    const uint32_t code = (uint32_t)(d >> 24) & 0xFFu;
    *score_raw = (int32_t)(uint32_t)(d >> 32);
    *status_raw = code == 0 ? 0 : code == 255u ? WFA_INTERNAL_FALLBACK : WFA_STATUS_MAX_STEPS_REACHED;
    return true;
  };
  auto start = [&]() -> bool {
    // (idle time of an instance: long enough for the next call of a loop of single alignments, short enough that nothing lingers)
    mb->idle_ticks = (uint32_t)std::max(1, knob(al, K_MAILBOX_IDLE_US, 2000)) * 100u;
    __atomic_store_n(&mb->quit, 0u, __ATOMIC_RELEASE);
    __atomic_store_n(&mb->alive, 1u, __ATOMIC_RELEASE);
    if (wfa::launch_slim_mailbox(want, full, adapt, al->mb_stream, al->mb_d) != 0) {
      (void)hipGetLastError();
      __atomic_store_n(&mb->alive, 0u, __ATOMIC_RELEASE);
      return false;
    }
    al->mb_args = want; al->mb_args_valid = true;
    return true;
  };
  auto give_up = [&]() {
    mailbox_quit(al);
    __atomic_store_n(&mb->done, (unsigned long long)k | (255ull << 24), __ATOMIC_RELEASE);   // (no instance is running: the request counts as consumed)
    ++al->mb_failures;
    return 0;
  };
  if (__atomic_load_n(&mb->alive, __ATOMIC_ACQUIRE) == 0 && !start()) return give_up();
  const double t0 = now_ms();
  for (uint32_t spins = 1;; ++spins) {
    if (answered()) return 1;
    if ((spins & 255u) == 0) {
      if (__atomic_load_n(&mb->alive, __ATOMIC_ACQUIRE) == 0) {
        // the instance left (idle time over) as the request arrived: its results, if any, were stored before alive = 0
        if (answered()) return 1;
        if (!start()) return give_up();
      }
      if (now_ms() - t0 > 100.0) return give_up();   // (something is wrong: the launch-per-call path takes the pair)
    }
    __builtin_ia32_pause();
  }
}

static int align_tiny(wfa_hip_aligner* al, int64_t n, const uint8_t* seqs, const int64_t* p_off, const int32_t* p_len,
                      const int64_t* t_off, const int32_t* t_len, int32_t* score, int32_t* status, uint8_t* cigar_ops,
                      const int64_t* cigar_off, int64_t* cigar_begin, int32_t* cigar_len) {
  const wfa_hip_config_t& c = al->cfg;
  const bool full = c.scope == WFA_SCOPE_FULL;
  if (n < 1 || n > TINY_MAX_PAIRS_BAND || knob(al, K_NO_TINY, 0)) return 0;
  if (c.memory_mode == WFA_MEM_BIWFA && (full || c.max_steps > 0 || c.heuristic != WFA_HEUR_NONE)) return 0;   // (the BiWFA kernel: the batch path)
  if (!score || !status || (full && cigar_ops && (!cigar_off || !cigar_begin || !cigar_len))) return 0;
  int64_t blob = 0, ops_total = 0;
  int max_len = 0, max_width = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int pl = p_len[i], tl = t_len[i];
    if (pl < 0 || tl < 0 || p_off[i] < 0 || t_off[i] < 0) { al->err = "negative length or offset"; return WFA_HIP_EINVAL; }
    if (c.span == WFA_SPAN_ENDSFREE && (c.pattern_begin_free > pl || c.pattern_end_free > pl || c.text_begin_free > tl || c.text_end_free > tl)) {
      al->err = "Ends-free parameters must be not larger than the sequences"; return WFA_HIP_EINVAL;
    }
    blob += (int64_t)((pl + 15) & ~15) + ((tl + 15) & ~15);
    ops_total += (int64_t)pl + tl;
    max_len = std::max(max_len, std::max(pl, tl)); max_width = std::max(max_width, pl + tl + 3);
  }
  if (max_len > 1000) return 0;   // long reads: the staged path (banded / wide kernels) is several times faster than this one's general kernel
  // input region: meta, byte offsets, op-region starts, the bytes; output region: results and op bytes
  const size_t o_meta = 0, o_pb = o_meta + (size_t)n * sizeof(WfaPairMeta), o_tb = o_pb + (size_t)n * 8, o_co = o_tb + (size_t)n * 8,
               o_blob = (o_co + (size_t)(n + 1) * 8 + 15) & ~(size_t)15, in_bytes = (o_blob + (size_t)blob + 15) & ~(size_t)15;
  const size_t o_score = TINY_IN_BYTES, o_status = o_score + (size_t)n * 4, o_cb = (o_status + (size_t)n * 4 + 7) & ~(size_t)7, o_cl = o_cb + (size_t)n * 8,
               o_ops = (o_cl + (size_t)n * 4 + 15) & ~(size_t)15;
  if (full && o_ops + (size_t)ops_total + 16 > TINY_BLOCK_BYTES) return 0;
  const bool general_fits = n <= TINY_MAX_PAIRS && in_bytes <= TINY_IN_BYTES;
  const bool band_form = !al->dcfg.lin && wfa::band_supported(al->dcfg, al->ncomp) && c.wildcard < 0 && knob(al, K_NO_TINY_BAND, 0) == 0 && knob(al, K_NO_BAND, 0) == 0;   // (lin: the banded kernel has no one-component form)
  if (!general_fits && !band_form) return 0;
  HIP_TRY(al, hipSetDevice(al->device));
  if (!al->tiny_h) {
    HIP_TRY(al, hipHostMalloc((void**)&al->tiny_h, TINY_BLOCK_BYTES, hipHostMallocMapped));
    HIP_TRY(al, hipMalloc((void**)&al->tiny_d, TINY_IN_BYTES));
  }
  uint8_t* h = al->tiny_h;
  if (!al->tiny_hd) HIP_TRY(al, hipHostGetDevicePointer((void**)&al->tiny_hd, h, 0));
  uint8_t* hd = al->tiny_hd;   // the pinned block as the device sees it
  // ---- gap-affine / gap-affine-2p with an instantiated penalty shape, pure ACGT: ONE launch of the banded kernel (one wave per
  // pair, 128 diagonals in registers).  The host packs the sequences to 2 bits into the pinned block; the kernel stages them
  // in LDS straight from there (no copy kernel), aligns, walks back in-kernel and writes results and op bytes into the pinned
  // block.  A pair the window cannot hold shows as status WFA_INTERNAL_FALLBACK and the call goes on to the general kernel below.
  if (band_form) {
    const size_t b_meta = 0, b_co = b_meta + (size_t)n * sizeof(WfaPairMeta), b_done = b_co + (size_t)(n + 1) * 8,
                 b_words = (b_done + (size_t)n * 4 + 15) & ~(size_t)15;
    WfaPairMeta* bm = reinterpret_cast<WfaPairMeta*>(h + b_meta);
    int64_t* bco = reinterpret_cast<int64_t*>(h + b_co);
    uint32_t* bw = reinterpret_cast<uint32_t*>(h + b_words);
    uint32_t w = 0;
    int64_t oo = 0;
    bool bad = false;
    size_t need_bytes = b_words;
    for (int64_t i = 0; i < n; ++i) need_bytes += (size_t)(((p_len[i] + 15) >> 4) + ((t_len[i] + 15) >> 4)) * 4;
    if (need_bytes + 64 <= TINY_IN_BYTES) {
      for (int64_t i = 0; i < n && !bad; ++i) {
        const int pl = p_len[i], tl = t_len[i];
        bm[i].plen = pl; bm[i].tlen = tl;
        bm[i].p_woff = w; bad |= wfa::host_pack_seq(seqs + p_off[i], pl, bw + w, -1); w += (uint32_t)((pl + 15) >> 4);
        bm[i].t_woff = w; bad |= wfa::host_pack_seq(seqs + t_off[i], tl, bw + w, -1); w += (uint32_t)((tl + 15) >> 4);
        bco[i] = oo; oo += (int64_t)pl + tl;
      }
      bco[n] = oo;
      for (int j = 0; j < 4; ++j) bw[w + j] = 0;   // (the funnel shift reads one word ahead)
    } else {
      bad = true;
    }
    if (!bad) {
      const int nch = 2;
      wfa::BandArgs ba;
      memset(&ba, 0, sizeof(ba));
      ba.words = reinterpret_cast<const uint32_t*>(hd + b_words); ba.meta = reinterpret_cast<const WfaPairMeta*>(hd + b_meta);
      ba.nwork = (uint32_t)n;
      ba.score = reinterpret_cast<int32_t*>(hd + o_score); ba.status = reinterpret_cast<int32_t*>(hd + o_status);
      ba.cigar_ops = hd + o_ops; ba.cigar_off = reinterpret_cast<const int64_t*>(hd + b_co);
      ba.cigar_begin = reinterpret_cast<int64_t*>(hd + o_cb); ba.cigar_len = reinterpret_cast<int32_t*>(hd + o_cl);
      ba.g = wfa::band_gcd(al->dcfg, al->ncomp == 5);
      ba.x = al->dcfg.x; ba.oe = al->dcfg.o1 + al->dcfg.e1; ba.e = al->dcfg.e1;
      if (al->ncomp == 5) { ba.oe2 = al->dcfg.o2 + al->dcfg.e2; ba.e2 = al->dcfg.e2; }
      ba.min_wf_len = al->dcfg.min_wf_len; ba.max_dist_thr = al->dcfg.max_dist_thr; ba.steps_between = al->dcfg.steps_between;
      ba.heur = al->dcfg.heuristic; ba.xdrop = al->dcfg.xdrop; ba.max_steps = al->dcfg.max_steps; ba.scope = al->dcfg.scope;
      ba.lds_words = ((max_len + 15) >> 4) + 4;
      // one pair, gap-affine: the resident kernel takes it from its mailbox when it can (mailbox_call) — its instance outlives the call,
      // so everything in its arguments is sized for the longest pair of this path, not for this one
      const bool try_mb = n == 1 && w + 4 <= 136 && al->ncomp == 3 && knob(al, K_BAND_SLIM, 1) != 0 && knob(al, K_NO_TINY_INLINE, 0) == 0 &&
                          knob(al, K_MAILBOX, 1) != 0 && knob(al, K_NO_TINY_POLL, 0) == 0 && al->mb_failures < 3;
      if (try_mb) ba.lds_words = ((1000 + 15) >> 4) + 5;
      ba.h16 = 1;
      ba.slim = knob(al, K_BAND_SLIM, 1);   // (wf-adaptive or no heuristic: wfa_slim_kernel — a third of the instructions per score step)
      ba.ef = (al->dcfg.endsfree && (al->dcfg.pbf | al->dcfg.pef | al->dcfg.tbf | al->dcfg.tef)) ? 1 : 0;
      ba.pbf = al->dcfg.pbf; ba.pef = al->dcfg.pef; ba.tbf = al->dcfg.tbf; ba.tef = al->dcfg.tef;
      if (full) {
        const int rec = ((al->ncomp != 5) ? 2 : 4) * 64 * nch;
        const long long records = std::max<long long>(256, (long long)((try_mb ? 1000 : max_len) * 0.9 * penalty_scale(al->dcfg)) / ba.g + 64);
        ba.hist_stride = ((int64_t)records * rec + 63) & ~63ll;
        // (one history slice per pair — 256 KB for 150 bp gap-affine — stays allocated for the aligner's lifetime: calls
        // whose slices exceed 256 MB or the free memory take the batch path, and so does a call whose workspace cannot be had)
        const size_t hist_bytes = (size_t)n * (size_t)ba.hist_stride * 4;
        if (hist_bytes > ((size_t)256 << 20) || (int64_t)hist_bytes > free_budget(al)) return 0;
        if (ensure_ws(al, hist_bytes) != WFA_HIP_OK) return 0;
        ba.hist = al->ws;
      }
      hipStream_t stream = al->stream;
      if (al->ws_event_recorded && al->ws_last_stream != stream) HIP_TRY(al, hipStreamWaitEvent(stream, al->ws_event, 0));
      // (a status that is none of the kernel's: a wave that wrote nothing would be seen)
      int32_t* hst0 = reinterpret_cast<int32_t*>(h + o_status);
      for (int64_t i = 0; i < n; ++i) hst0[i] = WFA_INTERNAL_FALLBACK;
      // completion: a flag per pair in the pinned block, stored by the kernel at system scope after the pair's results;
      // the host polls the flags (a few microseconds after the last store) and only then falls back on the stream wait
      volatile int32_t* hdone = reinterpret_cast<volatile int32_t*>(h + b_done);
      const bool poll = knob(al, K_NO_TINY_POLL, 0) == 0;
      for (int64_t i = 0; i < n; ++i) hdone[i] = 0;
      ba.done = poll ? reinterpret_cast<int32_t*>(hd + b_done) : nullptr;
      // one pair: lengths, op-region offsets and packed words ride in the kernel arguments (wfa_slim_kernel_one)
      uint32_t one[8 + 136];
      if (n == 1 && w + 4 <= 136 && ba.slim && knob(al, K_NO_TINY_INLINE, 0) == 0) {
        memcpy(one, &bm[0], 16); memcpy(one + 4, &bco[0], 16);
        memcpy(one + 8, bw, (size_t)(w + 4) * 4);
        ba.one = one;
      }
      bool seen = false;
      if (try_mb && ba.one) {
        // (a batch run still in flight may be using the workspace this pair's history goes to)
        if (full && al->ws_event_recorded) HIP_TRY(al, hipEventSynchronize(al->ws_event));
        wfa::BandArgs mba = ba;
        mba.done = nullptr;
        int32_t sr = 0, st = WFA_INTERNAL_FALLBACK;
        seen = mailbox_call(al, mba, full, al->dcfg.heuristic != WFA_HEUR_NONE, one, (size_t)8 + w + 4, &sr, &st) == 1;
        if (seen) {   // (the answer came in the mailbox's own word: the pinned block's score / status are what the launch path reads)
          std::atomic_thread_fence(std::memory_order_acquire);
          reinterpret_cast<int32_t*>(h + o_score)[0] = sr; hst0[0] = st;
        }
      }
      const unsigned rtc_failures = wfa::rtc_failure_count();
      if (!seen && wfa::launch_band(ba, nch, full, al->dcfg.heuristic != WFA_HEUR_NONE, true, (long long)n, stream) != 0) {
        if (al->dcfg.rtc && wfa::rtc_failure_count() != rtc_failures) {
          // a run-time shape that cannot be built: this aligner goes on without them (the batch path and the general kernel take the call)
          (void)hipGetLastError();
          al->dcfg.rtc = 0;
          al->rtc_note = std::string("run-time kernels switched off for this aligner: ") + wfa::rtc_last_error();
          return 0;
        }
        al->err = "band kernel launch failed"; return WFA_HIP_EDEVICE;
      }
      // (no event for the workspace: the call returns only after every pair's flag — stored behind its walk, the history's last
      // reader — or after the stream has drained)
      const bool served = seen;
      if (poll && !served) {
        const double t_poll = now_ms();
        for (;;) {
          int64_t got = 0;
          for (int64_t i = 0; i < n; ++i) got += (hdone[i] != 0);
          if (got == n) { seen = true; break; }
          if (now_ms() - t_poll > 20.0) break;   // (something is wrong or slow: the stream wait below decides)
          __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      if (!seen) HIP_TRY(al, hipStreamSynchronize(stream));
      const int32_t* hs = reinterpret_cast<const int32_t*>(h + o_score);
      bool handed_on = false;
      for (int64_t i = 0; i < n; ++i) handed_on |= (hst0[i] == WFA_INTERNAL_FALLBACK);
      if (!handed_on) {
        const int64_t* hcb = reinterpret_cast<const int64_t*>(h + o_cb); const int32_t* hcl = reinterpret_cast<const int32_t*>(h + o_cl);
        for (int64_t i = 0; i < n; ++i) {
          score[i] = tiny_score(al->dcfg, hs[i], hst0[i], p_len[i], t_len[i]); status[i] = hst0[i];
          if (cigar_len) cigar_len[i] = full ? hcl[i] : 0;
          if (cigar_begin) cigar_begin[i] = 0;
          if (full && cigar_ops) {
            const int64_t rel = hcb[i] - bco[i];
            if (hcl[i] > 0) memcpy(cigar_ops + cigar_off[i] + rel, h + o_ops + hcb[i], (size_t)hcl[i]);
            cigar_begin[i] = cigar_off[i] + rel;
          }
        }
        return 1;
      }
    }
  }
  if (!general_fits) return 0;   // (more pairs than the general-kernel form takes: the batch path)
  WfaPairMeta* meta = reinterpret_cast<WfaPairMeta*>(h + o_meta);
  int64_t* pb = reinterpret_cast<int64_t*>(h + o_pb); int64_t* tb = reinterpret_cast<int64_t*>(h + o_tb); int64_t* co = reinterpret_cast<int64_t*>(h + o_co);
  int64_t bo = 0, oo = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int pl = p_len[i], tl = t_len[i];
    meta[i].p_woff = 0; meta[i].t_woff = 0; meta[i].plen = pl; meta[i].tlen = tl;
    pb[i] = bo; memcpy(h + o_blob + bo, seqs + p_off[i], (size_t)pl); bo += (pl + 15) & ~15;
    tb[i] = bo; memcpy(h + o_blob + bo, seqs + t_off[i], (size_t)tl); bo += (tl + 15) & ~15;
    co[i] = oo; oo += (int64_t)pl + tl;
  }
  co[n] = oo;
  // geometry of the general kernel (plan_general / initial_arena_ints for these few pairs)
  int threads = (max_len > 2000 && c.heuristic != WFA_HEUR_ADAPTIVE) ? 256 : 64;
  threads = std::max(64, std::min(512, (knob(al, K_THREADS, threads) / 64) * 64));
  int64_t stride;
  if (full) {
    const int mi = 2 * al->ncomp + 4;
    stride = std::max<int64_t>((int64_t)max_len * (al->ncomp * 64 + mi) / 4 + (int64_t)max_width * al->ncomp * 4 + 4096 * mi, 1 << 14);
  } else {
    stride = (int64_t)al->dcfg.scope * al->ncomp * max_width;
  }
  stride = (stride + 63) & ~63ll;
  int rc = ensure_ws(al, (size_t)n * stride * 4);
  if (rc != WFA_HIP_OK) return rc;
  hipStream_t stream = al->stream;
  if (al->ws_event_recorded && al->ws_last_stream != stream) HIP_TRY(al, hipStreamWaitEvent(stream, al->ws_event, 0));
  const int n16 = (int)(in_bytes / 16);
  hipLaunchKernelGGL(wfa_tiny_copy_kernel, dim3(std::min(64, (n16 + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<uint4*>(al->tiny_d), reinterpret_cast<const uint4*>(hd), n16);
  WfaKernelArgs a;
  memset(&a, 0, sizeof(a));
  a.bytes = al->tiny_d + o_blob; a.meta = reinterpret_cast<const WfaPairMeta*>(al->tiny_d + o_meta);
  a.p_boff = reinterpret_cast<const int64_t*>(al->tiny_d + o_pb); a.t_boff = reinterpret_cast<const int64_t*>(al->tiny_d + o_tb);
  a.cigar_off = reinterpret_cast<const int64_t*>(al->tiny_d + o_co);
  a.nwork = (uint32_t)n;
  a.score = reinterpret_cast<int32_t*>(hd + o_score); a.status = reinterpret_cast<int32_t*>(hd + o_status);
  a.cigar_begin = reinterpret_cast<int64_t*>(hd + o_cb); a.cigar_len = reinterpret_cast<int32_t*>(hd + o_cl); a.cigar_ops = hd + o_ops;
  a.ws = al->ws; a.ws_stride = stride; a.cfg = al->gcfg;
  // (no overflow list: an arena that is too small shows as status WFA_INTERNAL_OVERFLOW below and the call takes the batch path)
  static uint32_t* const no_list = nullptr;
  a.fb_list = no_list; a.fb_count = nullptr;
  if (wfa::launch_general_any(al->gncomp, false, full, false, a, (int)n, threads, stream) != 0) { al->err = "general kernel launch failed"; return WFA_HIP_EDEVICE; }
  HIP_TRY(al, hipEventRecord(al->ws_event, stream));
  al->ws_event_recorded = true; al->ws_last_stream = stream;
  HIP_TRY(al, hipStreamSynchronize(stream));
  const int32_t* hs = reinterpret_cast<const int32_t*>(h + o_score); const int32_t* hst = reinterpret_cast<const int32_t*>(h + o_status);
  for (int64_t i = 0; i < n; ++i) if (hst[i] == WFA_STATUS_OOM && full) return 0;   // arena too small for this pair: the batch path grows it
  const int64_t* hcb = reinterpret_cast<const int64_t*>(h + o_cb); const int32_t* hcl = reinterpret_cast<const int32_t*>(h + o_cl);
  for (int64_t i = 0; i < n; ++i) {
    score[i] = tiny_score(al->gcfg, hs[i], hst[i], p_len[i], t_len[i]); status[i] = hst[i];
    if (cigar_len) cigar_len[i] = full ? hcl[i] : 0;
    if (cigar_begin) cigar_begin[i] = 0;
    if (full && cigar_ops) {
      const int64_t rel = hcb[i] - co[i];
      if (hcl[i] > 0) memcpy(cigar_ops + cigar_off[i] + rel, h + o_ops + hcb[i], (size_t)hcl[i]);
      cigar_begin[i] = cigar_off[i] + rel;
    }
  }
  return 1;
}

extern "C" int wfa_hip_align_batch(wfa_hip_aligner_t* al, int64_t n, const uint8_t* seqs,
                                   const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len,
                                   int32_t* score, int32_t* status, uint8_t* cigar_ops, const int64_t* cigar_off,
                                   int64_t* cigar_begin, int32_t* cigar_len) {
  if (!al) return WFA_HIP_EINVAL;
  if (n > 0 && n <= TINY_MAX_PAIRS_BAND && seqs && p_off && p_len && t_off && t_len) {
    const int trc = align_tiny(al, n, seqs, p_off, p_len, t_off, t_len, score, status, cigar_ops, cigar_off, cigar_begin, cigar_len);
    if (trc == 1) return WFA_HIP_OK;
    if (trc < 0) return trc;
  }
  const bool timing = al->knobs.set[K_TIMING];
  const double t0 = now_ms();
  // (the results call below synchronises the stream before this function returns, so the uploads need no wait of their own)
  wfa_hip_batch_t* b = batch_create_nosync(al, n, seqs, p_off, p_len, t_off, t_len);
  if (!b) return (al->err.find("failed:") != std::string::npos) ? WFA_HIP_EDEVICE : WFA_HIP_EINVAL;
  const double t1 = now_ms();
  int rc = wfa_hip_batch_run(b, nullptr);
  const double t2 = now_ms();
  if (rc == WFA_HIP_OK) rc = wfa_hip_batch_results(b, score, status, cigar_ops, cigar_off, cigar_begin, cigar_len);
  const double t3 = now_ms();
  wfa_hip_batch_destroy(b);
  if (timing) fprintf(stderr, "[wfa_hip] align_batch: create %.3f ms, enqueue %.3f ms, sync + results %.3f ms, destroy %.3f ms\n",
                      t1 - t0, t2 - t1, t3 - t2, now_ms() - t3);
  return rc;
}

extern "C" int wfa_hip_align_pair(wfa_hip_aligner_t* al, const uint8_t* pattern, int32_t plen, const uint8_t* text, int32_t tlen,
                                  int32_t* score, int32_t* status, uint8_t* cigar_ops, int64_t* cigar_begin, int32_t* cigar_len) {
  if (!al || !score || !status || plen < 0 || tlen < 0 || (plen > 0 && !pattern) || (tlen > 0 && !text)) return WFA_HIP_EINVAL;
  // one blob for the two sequences (a copy: the batch entry takes one blob + offsets, and the staged path behind it reads the blob
  // up to the last offset — offsets between two unrelated host objects would span whatever lies between them)
  std::vector<uint8_t>& blob = al->pair_blob;
  blob.resize((size_t)plen + (size_t)tlen + 16);
  if (plen > 0) memcpy(blob.data(), pattern, (size_t)plen);
  if (tlen > 0) memcpy(blob.data() + plen, text, (size_t)tlen);
  const uint8_t* base = blob.data();
  const int64_t p_off = 0, t_off = plen;
  const int64_t c_off[2] = {0, (int64_t)plen + tlen};
  int64_t cb = 0;
  int32_t cl = 0;
  const bool want = cigar_ops != nullptr;
  const int rc = wfa_hip_align_batch(al, 1, base, &p_off, &plen, &t_off, &tlen, score, status, cigar_ops, want ? c_off : nullptr,
                                     want ? &cb : nullptr, want ? &cl : nullptr);
  if (cigar_begin) *cigar_begin = cb;
  if (cigar_len) *cigar_len = cl;
  return rc;
}

static int64_t batch_extent(int64_t n, const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len, bool in2bit);
extern "C" int64_t wfa_hip_batch_extent(int64_t n, const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len) {
  return batch_extent(n, p_off, p_len, t_off, t_len, false);
}
extern "C" int64_t wfa_hip_batch_extent_packed2bits(int64_t n, const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len) {
  return batch_extent(n, p_off, p_len, t_off, t_len, true);
}
static int64_t batch_extent(int64_t n, const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len, bool in2bit) {
  if (n < 0 || (n > 0 && (!p_off || !p_len || !t_off || !t_len))) return -1;
  const int team = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, std::thread::hardware_concurrency()), n / 262144));
  std::vector<int64_t> ends((size_t)team, 0);
  auto work = [&](int t) {
    int64_t e = 0;
    for (int64_t i = n * t / team, hi = n * (t + 1) / team; i < hi; ++i) {
      if (p_off[i] < 0 || t_off[i] < 0 || p_len[i] < 0 || t_len[i] < 0) { e = -1; break; }
      e = in2bit ? std::max(e, std::max(p_off[i] + ((p_len[i] + 3) >> 2), t_off[i] + ((t_len[i] + 3) >> 2)))
                 : std::max(e, std::max(p_off[i] + p_len[i], t_off[i] + t_len[i]));
    }
    ends[(size_t)t] = e;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < team; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  int64_t e = 0;
  for (int64_t v : ends) { if (v < 0) return -1; e = std::max(e, v); }
  return e;
}

extern "C" int wfa_hip_pack_2bit(const uint8_t* seq, int32_t len, uint32_t* words, int form) {
  if (len < 0 || (len > 0 && (!seq || !words))) return WFA_HIP_EINVAL;
  return wfa::host_pack_seq(seq, len, words, form) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// several devices of one node: contiguous shards balanced by sum(plen + tlen), one host thread +
// aligner + stream per device, disjoint output slices, no collective
// ------------------------------------------------------------------------------------------------
extern "C" int wfa_hip_plan_shards(int64_t n, const int32_t* p_len, const int32_t* t_len, int nshards, int64_t* shard_begin) {
  if (n < 0 || nshards < 1 || !shard_begin || (n > 0 && (!p_len || !t_len))) return WFA_HIP_EINVAL;
  // work of a pair ~ its bases (+ a constant so that empty pairs still count); boundary s at the first pair whose prefix
  // reaches s / nshards of the total
  long double total = 0;
  for (int64_t i = 0; i < n; ++i) total += (long double)p_len[i] + t_len[i] + 16;
  shard_begin[0] = 0;
  long double acc = 0;
  int s = 1;
  for (int64_t i = 0; i < n && s < nshards; ++i) {
    acc += (long double)p_len[i] + t_len[i] + 16;
    while (s < nshards && acc >= total * s / nshards) shard_begin[s++] = i + 1;
  }
  while (s <= nshards) shard_begin[s++] = n;
  return WFA_HIP_OK;
}

struct wfa_hip_multi {
  std::vector<wfa_hip_aligner*> al;
  std::string err;
};

extern "C" wfa_hip_multi_t* wfa_hip_multi_create(const wfa_hip_config_t* cfg, const int* devices, int ndevices) {
  if (!cfg || !devices || ndevices < 1) { g_error = "invalid multi-device arguments"; return nullptr; }
  wfa_hip_multi* m = new wfa_hip_multi();
  for (int i = 0; i < ndevices; ++i) {
    wfa_hip_aligner* a = wfa_hip_create(cfg, devices[i]);
    if (!a) { for (wfa_hip_aligner* x : m->al) wfa_hip_destroy(x); delete m; return nullptr; }
    a->host_share *= ndevices;   // (the devices of this handle share the host's cores: each pipeline takes its part)
    m->al.push_back(a);
  }
  return m;
}

extern "C" void wfa_hip_multi_destroy(wfa_hip_multi_t* m) {
  if (!m) return;
  for (wfa_hip_aligner* x : m->al) wfa_hip_destroy(x);
  delete m;
}

extern "C" int wfa_hip_multi_set_config(wfa_hip_multi_t* m, const wfa_hip_config_t* cfg) {
  if (!m) return WFA_HIP_EINVAL;
  for (wfa_hip_aligner* x : m->al) {
    const int rc = wfa_hip_set_config(x, cfg);
    if (rc != WFA_HIP_OK) { m->err = x->err; return rc; }
  }
  return WFA_HIP_OK;
}

extern "C" const char* wfa_hip_multi_last_error(const wfa_hip_multi_t* m) { return m ? m->err.c_str() : g_error.c_str(); }

extern "C" int wfa_hip_multi_align_batch(wfa_hip_multi_t* m, int64_t n, const uint8_t* seqs,
                                         const int64_t* p_off, const int32_t* p_len, const int64_t* t_off, const int32_t* t_len,
                                         int32_t* score, int32_t* status, uint8_t* cigar_ops, const int64_t* cigar_off,
                                         int64_t* cigar_begin, int32_t* cigar_len) {
  if (!m || n < 0) return WFA_HIP_EINVAL;
  const int nd = (int)m->al.size();
  std::vector<int64_t> sb((size_t)nd + 1);
  if (wfa_hip_plan_shards(n, p_len, t_len, nd, sb.data()) != WFA_HIP_OK) { m->err = "invalid batch arguments"; return WFA_HIP_EINVAL; }
  std::vector<int> rcs((size_t)nd, WFA_HIP_OK);
  auto work = [&](int d) {
    const int64_t lo = sb[(size_t)d], cnt = sb[(size_t)d + 1] - lo;
    if (cnt == 0) return;
    rcs[(size_t)d] = wfa_hip_align_batch(m->al[(size_t)d], cnt, seqs, p_off + lo, p_len + lo, t_off + lo, t_len + lo, score + lo, status + lo,
                                         cigar_ops, cigar_off ? cigar_off + lo : nullptr, cigar_begin ? cigar_begin + lo : nullptr,
                                         cigar_len ? cigar_len + lo : nullptr);
  };
  std::vector<std::thread> th;
  for (int d = 1; d < nd; ++d) th.emplace_back(work, d);
  work(0);
  for (auto& x : th) x.join();
  for (int d = 0; d < nd; ++d)
    if (rcs[(size_t)d] != WFA_HIP_OK) { m->err = "device " + std::to_string(m->al[(size_t)d]->device) + ": " + m->al[(size_t)d]->err; return rcs[(size_t)d]; }
  return WFA_HIP_OK;
}
