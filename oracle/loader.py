"""ctypes loaders for the two CPU checkers (test infrastructure, see oracle/__init__.py).

* ``oracle()``  -> liboracle.so, the plain-C restatement (oracle/wfa_oracle.c)
* ``reference()`` -> oracle/_ref/libwfa_ref*.so, the real WFA2-lib v2.3 compiled from
  /root/reference by oracle/Makefile (absent sources => the prebuilt file is used).

Both expose the same batch signature as ``wfa_hip_align_batch`` minus the handle.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

_I32 = np.int32
_I64 = np.int64


class Config(ctypes.Structure):
    """Mirror of wfa_hip_config_t (include/wfa_hip.h)."""

    _fields_ = [(n, ctypes.c_int32) for n in (
        "distance", "match", "mismatch", "gap_opening", "gap_extension", "gap_opening2",
        "gap_extension2", "scope", "span", "pattern_begin_free", "pattern_end_free",
        "text_begin_free", "text_end_free", "heuristic", "min_wavefront_length",
        "max_distance_threshold", "steps_between_cutoffs", "xdrop", "memory_mode", "max_steps",
        "wildcard", "reserved")]


DIST = {"indel": 0, "levenshtein": 1, "linear": 2, "affine": 3, "affine2p": 4}
HEUR = {None: 0, "adaptive": 1, "X-drop": 2}
MEM = {"high": 0, "medium": 1, "low": 2, "biwfa": 3}


def make_config(distance="affine", match=0, mismatch=4, gap_opening=6, gap_extension=2,
                gap_opening2=24, gap_extension2=1, scope="full", span="ends-free",
                pattern_begin_free=0, pattern_end_free=0, text_begin_free=0, text_end_free=0,
                heuristic=None, min_wavefront_length=10, max_distance_threshold=50,
                steps_between_cutoffs=1, xdrop=20, memory_mode="high", max_steps=0,
                wildcard=None):
    """kwargs of pywfa.WavefrontAligner.__init__ (align.pyx:309-334) -> Config."""
    c = Config()
    c.distance = DIST[distance]
    c.match, c.mismatch = match, mismatch
    c.gap_opening, c.gap_extension = gap_opening, gap_extension
    c.gap_opening2, c.gap_extension2 = gap_opening2, gap_extension2
    c.scope = {"score": 0, "full": 1}[scope]
    c.span = {"end-to-end": 0, "ends-free": 1}[span]
    c.pattern_begin_free, c.pattern_end_free = pattern_begin_free, pattern_end_free
    c.text_begin_free, c.text_end_free = text_begin_free, text_end_free
    c.heuristic = HEUR[heuristic]
    c.min_wavefront_length = min_wavefront_length
    c.max_distance_threshold = max_distance_threshold
    c.steps_between_cutoffs = steps_between_cutoffs
    c.xdrop = xdrop
    c.memory_mode = MEM[memory_mode]
    c.max_steps = max_steps
    c.wildcard = -1 if wildcard is None else ord(wildcard.upper())
    c.reserved = 0
    return c


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference is present, oracle/_ref."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _cpu_has_v3():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = set(line.split(":", 1)[1].split())
                    return {"avx2", "bmi2", "fma", "movbe"} <= fl and ("abm" in fl or "lzcnt" in fl)
    except OSError:
        pass
    return False


def _bind(lib, name):
    fn = getattr(lib, name)
    p = ctypes.c_void_p
    fn.argtypes = [ctypes.POINTER(Config), ctypes.c_int64] + [p] * 11
    fn.restype = ctypes.c_int
    return fn


_cache = {}


def oracle():
    if "oracle" not in _cache:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _cache["oracle"] = _bind(ctypes.CDLL(path), "wfa_oracle_align_batch")
    return _cache["oracle"]


def reference_path():
    names = ["libwfa_ref_v3.so", "libwfa_ref.so"] if _cpu_has_v3() else ["libwfa_ref.so"]
    for n in names:
        p = os.path.join(HERE, "_ref", n)
        if os.path.exists(p):
            return p
    return None


def have_reference():
    return reference_path() is not None


def reference():
    if "ref" not in _cache:
        path = reference_path()
        if path is None:
            raise FileNotFoundError("oracle/_ref not built (needs /root/reference; run make -C oracle)")
        _cache["ref"] = _bind(ctypes.CDLL(path), "ref_align_batch")
    return _cache["ref"]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def run(fn, cfg, batch, want_cigar=None):
    """Run a checker on a batch dict (seqs,p_off,p_len,t_off,t_len as numpy arrays).

    Returns dict(score, status, cigars=list[bytes] or None).
    """
    n = len(batch["p_len"])
    seqs = np.ascontiguousarray(batch["seqs"], dtype=np.uint8)
    p_off = np.ascontiguousarray(batch["p_off"], dtype=_I64)
    t_off = np.ascontiguousarray(batch["t_off"], dtype=_I64)
    p_len = np.ascontiguousarray(batch["p_len"], dtype=_I32)
    t_len = np.ascontiguousarray(batch["t_len"], dtype=_I32)
    score = np.zeros(n, _I32)
    status = np.zeros(n, _I32)
    if want_cigar is None:
        want_cigar = cfg.scope == 1
    if want_cigar:
        cap = p_len.astype(_I64) + t_len.astype(_I64)
        cigar_off = np.zeros(n + 1, _I64)
        np.cumsum(cap, out=cigar_off[1:])
        ops = np.zeros(max(int(cigar_off[-1]), 1), np.uint8)
        cbeg = np.zeros(n, _I64)
        clen = np.zeros(n, _I32)
    else:
        cigar_off = ops = cbeg = clen = None
    rc = fn(ctypes.byref(cfg), n, _ptr(seqs), _ptr(p_off), _ptr(p_len), _ptr(t_off), _ptr(t_len),
            _ptr(score), _ptr(status), _ptr(ops), _ptr(cigar_off), _ptr(cbeg), _ptr(clen))
    if rc != 0:
        raise RuntimeError(f"checker returned {rc}")
    cigars = None
    if want_cigar:
        cigars = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in range(n)]
    return {"score": score, "status": status, "cigars": cigars}


def check_cigars(cfg, batch, score, ops, cbeg, clen, check_score=True):
    """Property check of GPU transcripts (see wfa_oracle_check_cigars). Returns (n_bad, first_bad)."""
    oracle()
    lib = ctypes.CDLL(os.path.join(HERE, "liboracle.so"))
    fn = lib.wfa_oracle_check_cigars
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.POINTER(Config), ctypes.c_int64] + [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_void_p]
    first = ctypes.c_int64(-1)
    seqs = np.ascontiguousarray(batch["seqs"], dtype=np.uint8)
    arrs = [np.ascontiguousarray(batch["p_off"], np.int64), np.ascontiguousarray(batch["p_len"], np.int32),
            np.ascontiguousarray(batch["t_off"], np.int64), np.ascontiguousarray(batch["t_len"], np.int32),
            np.ascontiguousarray(score, np.int32), np.ascontiguousarray(ops, np.uint8),
            np.ascontiguousarray(cbeg, np.int64), np.ascontiguousarray(clen, np.int32)]
    bad = fn(ctypes.byref(cfg), len(arrs[1]), _ptr(seqs), *[_ptr(a) for a in arrs], 1 if check_score else 0,
             ctypes.byref(first))
    return int(bad), int(first.value)


def reference_mt(cfg, batch, nthreads, repeat=1):
    """Scores / statuses of the real reference on `nthreads` host threads (one aligner per thread)."""
    reference()
    lib = ctypes.CDLL(reference_path())
    fn = lib.ref_align_batch_mt
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(Config), ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 7
    n = len(batch["p_len"])
    seqs = np.ascontiguousarray(batch["seqs"], dtype=np.uint8)
    arrs = [np.ascontiguousarray(batch["p_off"], np.int64), np.ascontiguousarray(batch["p_len"], np.int32),
            np.ascontiguousarray(batch["t_off"], np.int64), np.ascontiguousarray(batch["t_len"], np.int32)]
    score = np.zeros(n, np.int32)
    status = np.zeros(n, np.int32)
    rc = fn(ctypes.byref(cfg), int(nthreads), int(repeat), n, _ptr(seqs), *[_ptr(a) for a in arrs], _ptr(score), _ptr(status))
    if rc != 0:
        raise RuntimeError(f"reference returned {rc}")
    return {"score": score, "status": status, "cigars": None}


def reference_mt_full(cfg, batch, nthreads=None, want_cigar=None):
    """`run(reference(), ...)` on several host threads (pairs handed out one at a time, one aligner per thread): scores,
    statuses and op strings of the real reference for the expensive configurations."""
    reference()
    lib = ctypes.CDLL(reference_path())
    fn = lib.ref_align_batch_mt_full
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(Config), ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 11
    nthreads = int(nthreads or os.cpu_count() or 1)
    n = len(batch["p_len"])
    seqs = np.ascontiguousarray(batch["seqs"], dtype=np.uint8)
    p_off = np.ascontiguousarray(batch["p_off"], np.int64); p_len = np.ascontiguousarray(batch["p_len"], np.int32)
    t_off = np.ascontiguousarray(batch["t_off"], np.int64); t_len = np.ascontiguousarray(batch["t_len"], np.int32)
    score = np.zeros(n, np.int32)
    status = np.zeros(n, np.int32)
    if want_cigar is None:
        want_cigar = cfg.scope == 1
    cigar_off = ops = cbeg = clen = None
    if want_cigar:
        cigar_off = np.zeros(n + 1, np.int64)
        np.cumsum(p_len.astype(np.int64) + t_len.astype(np.int64), out=cigar_off[1:])
        ops = np.zeros(max(int(cigar_off[-1]), 1), np.uint8)
        cbeg = np.zeros(n, np.int64)
        clen = np.zeros(n, np.int32)
    rc = fn(ctypes.byref(cfg), min(nthreads, max(n, 1)), n, _ptr(seqs), _ptr(p_off), _ptr(p_len), _ptr(t_off), _ptr(t_len),
            _ptr(score), _ptr(status), _ptr(ops), _ptr(cigar_off), _ptr(cbeg), _ptr(clen))
    if rc != 0:
        raise RuntimeError(f"reference returned {rc}")
    cigars = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in range(n)] if want_cigar else None
    return {"score": score, "status": status, "cigars": cigars}


def oracle_undefined_reads(reset=False):
    """How often the oracle met the one undefined branch of the reference's ends-free re-seeding (match < 0 with free begins,
    wavefront_compute.c:229-251: offsets[0] is read without ever being written) since the last reset.  The oracle reads NULL
    there; parity against the real library is only claimed where this stays 0."""
    oracle()
    lib = ctypes.CDLL(os.path.join(HERE, "liboracle.so"))
    lib.wfa_oracle_undefined_reads.argtypes = [ctypes.c_int]
    lib.wfa_oracle_undefined_reads.restype = ctypes.c_int64
    return int(lib.wfa_oracle_undefined_reads(1 if reset else 0))


def oracle_counters(reset=False):
    """(M offsets, offsets of all components, bases compared) the oracle has processed since the last reset
    (bench.py's "offsets/s" figure)."""
    oracle()
    lib = ctypes.CDLL(os.path.join(HERE, "liboracle.so"))
    lib.wfa_oracle_counters.argtypes = [ctypes.c_void_p]
    lib.wfa_oracle_counters.restype = None
    if reset:
        lib.wfa_oracle_counters(None)
        return (0, 0, 0)
    out = (ctypes.c_int64 * 3)()
    lib.wfa_oracle_counters(out)
    return (int(out[0]), int(out[1]), int(out[2]))
