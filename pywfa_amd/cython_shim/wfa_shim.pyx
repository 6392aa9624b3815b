# cython: language_level=3
# wfa_shim.pyx — the thin Cython host north_star describes: pywfa's WavefrontAligner surface for one pair per call
# (pywfa/align.pyx:306-467, 731-757) over the C ABI of libwfa_hip.so instead of WFA2-lib.  Just enough of the class to show the
# binding compiled and working (construct / wavefront_align / score / status / cigarstring / cigar_print_pretty); the full
# drop-in with batches, properties and AlignmentResult is pywfa_amd/align.py.
from libc.stdint cimport int32_t, int64_t, uint8_t
from wfa_hip cimport *

DIST = {"indel": 0, "levenshtein": 1, "linear": 2, "affine": 3, "affine2p": 4}


def abi_version():
    return wfa_hip_abi_version()


def device_count():
    return wfa_hip_device_count()


cdef class ShimAligner:
    cdef wfa_hip_aligner_t* aligner
    cdef wfa_hip_config_t cfg
    cdef bytes _bpattern
    cdef bytes _btext
    cdef bytearray _ops
    cdef int32_t _score, _status, _c_len
    cdef int64_t _c_begin

    def __cinit__(self):
        self.aligner = NULL

    def __init__(self, pattern=None, distance="affine", int match=0, int mismatch=4, int gap_opening=6, int gap_extension=2,
                 int gap_opening2=24, int gap_extension2=1, scope="full", span="ends-free", int device=0, create=True):
        wfa_hip_config_default(&self.cfg)                      # was: wavefront_aligner_attr_default (align.pyx:344)
        self.cfg.distance = DIST[distance]
        self.cfg.match = match; self.cfg.mismatch = mismatch
        self.cfg.gap_opening = gap_opening; self.cfg.gap_extension = gap_extension
        self.cfg.gap_opening2 = gap_opening2; self.cfg.gap_extension2 = gap_extension2
        self.cfg.scope = 1 if scope == "full" else 0
        self.cfg.span = 1 if span == "ends-free" else 0
        cdef char msg[256]
        if wfa_hip_config_validate(&self.cfg, msg, 256) != 0:   # the reference exit(1)s here (wavefront_penalties.c:101-112)
            raise ValueError(msg.decode())
        self._bpattern = pattern.upper().encode("ascii") if pattern is not None else b""
        self._ops = bytearray()
        self._score = 0; self._status = 0; self._c_len = 0; self._c_begin = 0
        if create:
            self.aligner = wfa_hip_create(&self.cfg, device)    # was: wavefront_aligner_new(&attributes) (align.pyx:419)
            if self.aligner == NULL:
                raise RuntimeError(wfa_hip_global_error().decode())

    def __dealloc__(self):
        if self.aligner != NULL:
            wfa_hip_destroy(self.aligner)                       # was: wavefront_aligner_delete (align.pyx:881-883)

    @property
    def config(self):
        return dict(distance=self.cfg.distance, match=self.cfg.match, mismatch=self.cfg.mismatch, gap_opening=self.cfg.gap_opening,
                    gap_extension=self.cfg.gap_extension, scope=self.cfg.scope, span=self.cfg.span, xdrop=self.cfg.xdrop,
                    min_wavefront_length=self.cfg.min_wavefront_length, wildcard=self.cfg.wildcard)

    def wavefront_align(self, text, pattern=None):
        """align.pyx:421-443: returns the score; the op string is kept for cigarstring."""
        if self.aligner == NULL:
            raise RuntimeError("no aligner (create=False)")
        if pattern is not None:
            self._bpattern = pattern.upper().encode("ascii")
        self._btext = text.upper().encode("ascii")
        cdef bytes blob = self._bpattern + self._btext
        cdef int64_t p_off = 0, t_off = len(self._bpattern)
        cdef int64_t c_off[2]
        cdef int32_t p_len = <int32_t>len(self._bpattern), t_len = <int32_t>len(self._btext)
        c_off[0] = 0; c_off[1] = p_len + t_len
        self._ops = bytearray(max(p_len + t_len, 1))
        cdef uint8_t* ops = self._ops
        cdef const uint8_t* seqs = <const uint8_t*>blob
        cdef int rc = wfa_hip_align_batch(self.aligner, 1, seqs, &p_off, &p_len, &t_off, &t_len,
                                          &self._score, &self._status, ops, c_off, &self._c_begin, &self._c_len)
        if rc != 0:
            raise RuntimeError(wfa_hip_last_error(self.aligner).decode())
        return self._score

    @property
    def score(self):
        return self._score

    @property
    def status(self):
        return self._status

    @property
    def cigarstring(self):
        """align.pyx:731-757: run-length encoding of operations[begin_offset:end_offset]."""
        ops = bytes(self._ops[self._c_begin:self._c_begin + self._c_len])
        out, i = [], 0
        while i < len(ops):
            j = i
            while j < len(ops) and ops[j] == ops[i]:
                j += 1
            out.append(f"{j - i}{chr(ops[i])}")
            i = j
        return "".join(out)


def sprint_pretty(bytes ops, bytes pattern, bytes text):
    """cigar_print_pretty (align.pyx:445-459) as a string: host only."""
    cdef const uint8_t* o = <const uint8_t*>ops
    cdef const uint8_t* p = <const uint8_t*>pattern
    cdef const uint8_t* t = <const uint8_t*>text
    cdef int64_t need = wfa_hip_cigar_sprint_pretty(o, len(ops), p, <int32_t>len(pattern), t, <int32_t>len(text), NULL, 0)
    if need < 0:
        raise ValueError("invalid arguments")
    buf = bytearray(need + 1)
    cdef char* b = buf
    wfa_hip_cigar_sprint_pretty(o, len(ops), p, <int32_t>len(pattern), t, <int32_t>len(text), b, need + 1)
    return bytes(buf[:need]).decode("ascii", "replace")
