# wfa_hip.pxd — the Cython declarations of include/wfa_hip.h a pywfa maintainer adds in place of pywfa/WFA_wrap.pxd
# (INTEGRATION.md §2).  tests/test_cython_shim.py compiles this file and wfa_shim.pyx against the header and libwfa_hip.so.
from libc.stdint cimport int32_t, int64_t, uint8_t

cdef extern from "wfa_hip.h" nogil:
    ctypedef struct wfa_hip_config_t:
        int32_t distance, match, mismatch, gap_opening, gap_extension, gap_opening2, gap_extension2
        int32_t scope, span, pattern_begin_free, pattern_end_free, text_begin_free, text_end_free
        int32_t heuristic, min_wavefront_length, max_distance_threshold, steps_between_cutoffs, xdrop
        int32_t memory_mode, max_steps, wildcard, reserved
    ctypedef struct wfa_hip_aligner_t
    int wfa_hip_abi_version()
    int wfa_hip_device_count()
    int wfa_hip_config_default(wfa_hip_config_t* cfg)
    int wfa_hip_config_validate(const wfa_hip_config_t* cfg, char* err, size_t errlen)
    wfa_hip_aligner_t* wfa_hip_create(const wfa_hip_config_t* cfg, int device)
    void wfa_hip_destroy(wfa_hip_aligner_t* aligner)
    int wfa_hip_set_config(wfa_hip_aligner_t* aligner, const wfa_hip_config_t* cfg)
    const char* wfa_hip_last_error(const wfa_hip_aligner_t* aligner)
    const char* wfa_hip_global_error()
    int wfa_hip_align_batch(wfa_hip_aligner_t* aligner, int64_t n, const uint8_t* seqs,
                            const int64_t* p_off, const int32_t* p_len,
                            const int64_t* t_off, const int32_t* t_len,
                            int32_t* score, int32_t* status, uint8_t* cigar_ops,
                            const int64_t* cigar_off, int64_t* cigar_begin, int32_t* cigar_len)
    int64_t wfa_hip_cigar_sprint_pretty(const uint8_t* ops, int64_t ops_len, const uint8_t* pattern, int32_t plen,
                            const uint8_t* text, int32_t tlen, char* out, int64_t cap)
