# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""_host.pyx — the compiled host of the drop-in (north_star: "the Cython host in align.pyx becomes a thin C-ABI shim that marshals
batches of (pattern, text) pairs").

pywfa's host is compiled Cython (/root/reference/pywfa/align.pyx:421-443: ``text.upper().encode("ascii")``, then one
``wavefront_align`` per call).  This module is its counterpart in front of ``libwfa_hip.so``:

* ``from_strings``: a list of texts against one pattern, or against a list of patterns, becomes the batch the C ABI takes
  (include/wfa_hip.h: one ASCII blob + offsets + lengths) — the same dict ``pywfa_amd.datagen.from_strings`` builds in Python,
  byte for byte, by OpenMP threads that read the ``str`` / ``bytes`` objects' buffers in place and upper-case on the way
  (``host_core.c``); the caller's thread keeps the GIL meanwhile, so nothing can move under them;
* ``PairCaller``: one pair per call (``WavefrontAligner.wavefront_align`` / ``__call__``) over ``wfa_hip_align_pair`` without
  ctypes marshalling.

The library's entry points are BOUND at run time (``bind``: addresses from the ctypes handle ``_native.lib()``), so this module and
``_native.py`` always talk to the same loaded library (``WFA_HIP_LIB`` included) and the extension links nothing.  Built in-tree
by ``__graft_entry__.build()`` (``pywfa_amd/host/build_host.py``: cython + gcc); ``pywfa_amd/align.py`` uses the ctypes / Python
path when the extension is absent.  No CPU alignment path here either: every pair goes to the GPU."""
from libc.stdint cimport int32_t, int64_t, uint8_t, uintptr_t
from cpython.bytes cimport PyBytes_AS_STRING, PyBytes_GET_SIZE
from cpython.ref cimport PyObject

import numpy as np

cdef extern from "Python.h":
    int PyUnicode_CheckExact(object o)
    int PyUnicode_IS_COMPACT_ASCII(object o)
    Py_ssize_t PyUnicode_GET_LENGTH(object o)
    void* PyUnicode_DATA(object o)
    PyObject** PySequence_Fast_ITEMS(object o)

cdef extern from "host_core.c":
    int64_t wfa_host_lengths(PyObject** patterns, PyObject** texts, int64_t n, int64_t* len64, int threads)
    void wfa_host_fill(PyObject** patterns, PyObject** texts, int64_t n, const int64_t* p_off, const int64_t* t_off, uint8_t* blob, int threads)

ctypedef int (*align_pair_fn)(void*, const uint8_t*, int32_t, const uint8_t*, int32_t, int32_t*, int32_t*, uint8_t*, int64_t*, int32_t*) noexcept nogil

cdef align_pair_fn c_align_pair = NULL
cdef int n_threads = 16


def bind(uintptr_t align_pair, int threads=16):
    """The address of wfa_hip_align_pair in the loaded library; threads for the batch marshalling."""
    global c_align_pair, n_threads
    c_align_pair = <align_pair_fn>align_pair
    n_threads = max(1, min(threads, 64))


def normalise(object s):
    """What the reference does to a sequence argument (align.pyx:432,435): ``s.upper().encode("ascii")``; bytes are taken as they are."""
    if isinstance(s, (bytes, bytearray)):
        return bytes(s)
    return s.upper().encode("ascii")


# ---------------------------------------------------------------------------------------------------------------- batches
cdef object scratch_array(dict scratch, str key, int64_t n, object dtype):
    """A NumPy array of >= n elements kept in `scratch` between calls (a fresh 300 MB blob per call is 70 k page faults)."""
    if scratch is None:
        return np.empty(max(n, 1), dtype)
    arr = scratch.get(key)
    if arr is None or arr.shape[0] < n:
        arr = np.empty(max(n + n // 8, 1), dtype)
        scratch[key] = arr
    return arr


def from_strings(object patterns, list texts, dict scratch=None):
    """The batch dict of ``datagen.from_strings(patterns, texts, upper=True)`` (same layout: a shared pattern first, otherwise
    pattern / text interleaved; 64 zero bytes behind), or None when an object is not an exact ASCII ``str`` / ``bytes`` (the
    Python path then converts it and raises the reference's errors).  `scratch`: a dict the caller keeps between calls — the arrays
    of the batch are then views of buffers kept in it (valid until the next call with the same dict)."""
    cdef int64_t n = len(texts), i
    cdef bint shared = isinstance(patterns, (str, bytes))
    cdef list plist
    cdef PyObject** pitems = NULL
    cdef PyObject** titems = PySequence_Fast_ITEMS(texts)
    cdef bytes head = b""
    if shared:
        head = normalise(patterns)
    else:
        plist = patterns
        if len(plist) != n:
            raise ValueError("patterns and texts differ in length")
        pitems = PySequence_Fast_ITEMS(plist)
    cdef int64_t step = 1 if shared else 2
    lens = scratch_array(scratch, "lens", step * n, np.int64)
    cdef int64_t[::1] lv = lens
    if n and wfa_host_lengths(pitems, titems, n, &lv[0], n_threads) != 0:
        return None
    offs = scratch_array(scratch, "offs", step * n, np.int64)
    cdef int64_t[::1] ov = offs
    cdef int64_t cur = len(head)
    for i in range(step * n):
        ov[i] = cur; cur += lv[i]
    blob = scratch_array(scratch, "blob", cur + 64, np.uint8)
    cdef uint8_t[::1] bv = blob
    cdef Py_ssize_t hn = len(head)
    cdef const uint8_t* hp = <const uint8_t*>PyBytes_AS_STRING(head)
    for i in range(hn):
        bv[i] = hp[i]
    for i in range(64):
        bv[cur + i] = 0
    lens = lens[:step * n]; offs = offs[:step * n]
    if shared:
        p_off = np.zeros(n, np.int64); p_len = np.full(n, hn, np.int32)
        t_off = offs; t_len = lens.astype(np.int32)
    else:
        p_off = np.ascontiguousarray(offs[0::2]); t_off = np.ascontiguousarray(offs[1::2])
        p_len = lens[0::2].astype(np.int32); t_len = lens[1::2].astype(np.int32)
    cdef int64_t[::1] pov = p_off
    cdef int64_t[::1] tov = t_off
    if n:
        wfa_host_fill(pitems, titems, n, &pov[0], &tov[0], &bv[0], n_threads)
    return {"seqs": blob[:cur + 64], "p_off": p_off, "p_len": p_len, "t_off": t_off, "t_len": t_len}


# ---------------------------------------------------------------------------------------------------------------- one pair
cdef class PairCaller:
    """``wfa_hip_align_pair`` with its out-parameters kept between calls (the loop of ``wavefront_align(text)`` calls pywfa is
    built around)."""
    cdef void* handle
    cdef int32_t score, status, clen
    cdef int64_t cbeg
    cdef bytearray ops

    def __init__(self, uintptr_t aligner_handle):
        self.handle = <void*>aligner_handle
        self.ops = bytearray(1024)

    def align(self, bytes pattern, bytes text, bint want_cigar):
        """-> (rc, score, status, op bytes or None); both sequences already normalised."""
        if c_align_pair == NULL:
            raise RuntimeError("pywfa_amd.host._host is not bound to the library")
        cdef Py_ssize_t pn = PyBytes_GET_SIZE(pattern), tn = PyBytes_GET_SIZE(text)
        cdef uint8_t* ops = NULL
        if want_cigar:
            if len(self.ops) < pn + tn + 1:
                self.ops = bytearray(2 * (pn + tn) + 64)
            ops = self.ops
        cdef int rc
        cdef const uint8_t* pp = <const uint8_t*>PyBytes_AS_STRING(pattern)
        cdef const uint8_t* tp = <const uint8_t*>PyBytes_AS_STRING(text)
        with nogil:
            rc = c_align_pair(self.handle, pp, <int32_t>pn, tp, <int32_t>tn, &self.score, &self.status, ops, &self.cbeg, &self.clen)
        if rc != 0:
            return rc, 0, 0, None
        if want_cigar:
            return 0, self.score, self.status, bytes(self.ops[self.cbeg:self.cbeg + self.clen])
        return 0, self.score, self.status, None
