"""ctypes binding of libwfa_hip.so (the C ABI declared in include/wfa_hip.h).

This is the reference-side binding a pywfa maintainer would add instead of ``WFA_wrap.pxd``
(INTEGRATION.md shows the Cython form).  There is NO CPU fallback: if the library cannot be loaded,
or no HIP device is present, every alignment entry point raises.
"""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WFA_HIP_LIB") or os.path.join(_HERE, "libwfa_hip.so")  # (WFA_HIP_LIB: development builds)

ABI_VERSION = 3

OK, EINVAL, ENOTSUP, EDEVICE = 0, -1, -2, -3

DIST = {"indel": 0, "levenshtein": 1, "linear": 2, "affine": 3, "affine2p": 4}
DIST_NAMES = {v: k for k, v in DIST.items()}
SCOPE = {"score": 0, "full": 1}
SPAN = {"end-to-end": 0, "ends-free": 1}
HEUR = {None: 0, "adaptive": 1, "X-drop": 2}
HEUR_NAMES = {v: k for k, v in HEUR.items()}
MEM = {"high": 0, "medium": 1, "low": 2, "biwfa": 3}
MEM_NAMES = {v: k for k, v in MEM.items()}


class Config(ctypes.Structure):
    """wfa_hip_config_t"""

    _fields_ = [(n, ctypes.c_int32) for n in (
        "distance", "match", "mismatch", "gap_opening", "gap_extension", "gap_opening2",
        "gap_extension2", "scope", "span", "pattern_begin_free", "pattern_end_free",
        "text_begin_free", "text_end_free", "heuristic", "min_wavefront_length",
        "max_distance_threshold", "steps_between_cutoffs", "xdrop", "memory_mode", "max_steps",
        "wildcard", "reserved")]

    def copy(self):
        c = Config()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(self), ctypes.sizeof(Config))
        return c


class NativeError(RuntimeError):
    pass


_lib = None

# every symbol include/wfa_hip.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "wfa_hip_abi_version", "wfa_hip_device_count", "wfa_hip_global_error", "wfa_hip_config_default",
    "wfa_hip_config_validate", "wfa_hip_create", "wfa_hip_destroy", "wfa_hip_set_config",
    "wfa_hip_get_config", "wfa_hip_last_error", "wfa_hip_align_batch", "wfa_hip_batch_create",
    "wfa_hip_batch_destroy", "wfa_hip_batch_run", "wfa_hip_batch_sync", "wfa_hip_batch_results",
    "wfa_hip_batch_last_kernel_ms", "wfa_hip_batch_algorithmic_bytes", "wfa_hip_batch_fallback_pairs",
    "wfa_hip_batch_rle_counts", "wfa_hip_batch_rle_runs",
    "wfa_hip_plan_shards", "wfa_hip_plan_host_threads", "wfa_hip_multi_create", "wfa_hip_multi_destroy", "wfa_hip_multi_set_config",
    "wfa_hip_multi_last_error", "wfa_hip_multi_align_batch", "wfa_hip_pack_2bit", "wfa_hip_batch_extent",
    "wfa_hip_align_batch_packed2bits", "wfa_hip_batch_create_packed2bits", "wfa_hip_cigar_sprint_pretty",
    "wfa_hip_batch_extent_packed2bits", "wfa_hip_align_pair", "wfa_hip_upload_info",
]


def lib():
    """Load libwfa_hip.so (built in-tree by ``__graft_entry__.build()`` / csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} is missing: build it with pywfa_amd/csrc/build.sh "
                          "(there is no CPU fallback)")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    cfgp = ctypes.POINTER(Config)
    L.wfa_hip_abi_version.restype = ctypes.c_int
    L.wfa_hip_device_count.restype = ctypes.c_int
    L.wfa_hip_global_error.restype = ctypes.c_char_p
    L.wfa_hip_config_default.argtypes = [cfgp]
    L.wfa_hip_config_validate.argtypes = [cfgp, ctypes.c_char_p, ctypes.c_size_t]
    L.wfa_hip_create.argtypes = [cfgp, ctypes.c_int]
    L.wfa_hip_create.restype = vp
    L.wfa_hip_destroy.argtypes = [vp]
    L.wfa_hip_destroy.restype = None
    L.wfa_hip_set_config.argtypes = [vp, cfgp]
    L.wfa_hip_get_config.argtypes = [vp, cfgp]
    L.wfa_hip_last_error.argtypes = [vp]
    L.wfa_hip_last_error.restype = ctypes.c_char_p
    L.wfa_hip_align_batch.argtypes = [vp, i64] + [vp] * 11
    L.wfa_hip_align_pair.argtypes = [vp, ctypes.c_char_p, i32, ctypes.c_char_p, i32, vp, vp, vp, vp, vp]
    L.wfa_hip_batch_create.argtypes = [vp, i64] + [vp] * 5
    L.wfa_hip_batch_create.restype = vp
    L.wfa_hip_align_batch_packed2bits.argtypes = [vp, i64] + [vp] * 11
    L.wfa_hip_batch_create_packed2bits.argtypes = [vp, i64] + [vp] * 5
    L.wfa_hip_batch_create_packed2bits.restype = vp
    L.wfa_hip_cigar_sprint_pretty.argtypes = [vp, i64, vp, i32, vp, i32, vp, i64]
    L.wfa_hip_cigar_sprint_pretty.restype = i64
    L.wfa_hip_batch_destroy.argtypes = [vp]
    L.wfa_hip_batch_destroy.restype = None
    L.wfa_hip_batch_run.argtypes = [vp, vp]
    L.wfa_hip_batch_sync.argtypes = [vp]
    L.wfa_hip_batch_results.argtypes = [vp] * 7
    L.wfa_hip_batch_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i64)]
    L.wfa_hip_batch_algorithmic_bytes.argtypes = [vp]
    L.wfa_hip_batch_algorithmic_bytes.restype = i64
    L.wfa_hip_batch_fallback_pairs.argtypes = [vp]
    L.wfa_hip_batch_fallback_pairs.restype = i64
    L.wfa_hip_batch_rle_counts.argtypes = [vp, vp, vp]
    L.wfa_hip_batch_rle_counts.restype = i64
    L.wfa_hip_batch_rle_runs.argtypes = [vp, vp, vp]
    L.wfa_hip_plan_shards.argtypes = [i64, vp, vp, ctypes.c_int, vp]
    L.wfa_hip_plan_host_threads.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.wfa_hip_pack_2bit.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int]
    L.wfa_hip_upload_info.argtypes = [vp, vp, ctypes.c_int]
    L.wfa_hip_batch_extent.argtypes = [i64, vp, vp, vp, vp]
    L.wfa_hip_batch_extent.restype = i64
    L.wfa_hip_batch_extent_packed2bits.argtypes = [i64, vp, vp, vp, vp]
    L.wfa_hip_batch_extent_packed2bits.restype = i64
    L.wfa_hip_multi_create.argtypes = [cfgp, vp, ctypes.c_int]
    L.wfa_hip_multi_create.restype = vp
    L.wfa_hip_multi_destroy.argtypes = [vp]
    L.wfa_hip_multi_destroy.restype = None
    L.wfa_hip_multi_set_config.argtypes = [vp, cfgp]
    L.wfa_hip_multi_last_error.argtypes = [vp]
    L.wfa_hip_multi_last_error.restype = ctypes.c_char_p
    L.wfa_hip_multi_align_batch.argtypes = [vp, i64] + [vp] * 11
    if L.wfa_hip_abi_version() != ABI_VERSION:
        raise NativeError("libwfa_hip.so ABI version mismatch: rebuild it")
    _lib = L
    return L



_HOST = False   # False = not looked for yet; None = absent


def compiled_host():
    """pywfa_amd.host._host (Cython, built in-tree by __graft_entry__.build()) bound to the loaded library, or None when the
    extension has not been built (WFA_HIP_NO_COMPILED_HOST=1: ignore it — the ctypes / Python host, for tests)."""
    global _HOST
    if _HOST is False:
        _HOST = None
        if os.environ.get("WFA_HIP_NO_COMPILED_HOST") != "1":
            try:
                from .host import _host
                threads = max(1, min(64, (os.cpu_count() or 2) // 2))
                _host.bind(ctypes.cast(lib().wfa_hip_align_pair, ctypes.c_void_p).value, threads)
                _HOST = _host
            except ImportError:
                _HOST = None
    return _HOST

def default_config():
    c = Config()
    lib().wfa_hip_config_default(ctypes.byref(c))
    return c


def validate(cfg):
    """Return (code, message)."""
    buf = ctypes.create_string_buffer(256)
    rc = lib().wfa_hip_config_validate(ctypes.byref(cfg), buf, 256)
    return rc, buf.value.decode()


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _check_batch(batch):
    if "packed" in batch:
        return _check_packed(batch)
    seqs = np.ascontiguousarray(batch["seqs"], dtype=np.uint8)
    p_off = np.ascontiguousarray(batch["p_off"], dtype=np.int64)
    t_off = np.ascontiguousarray(batch["t_off"], dtype=np.int64)
    p_len = np.ascontiguousarray(batch["p_len"], dtype=np.int32)
    t_len = np.ascontiguousarray(batch["t_len"], dtype=np.int32)
    n = p_len.shape[0]
    if not (p_off.shape[0] == t_off.shape[0] == t_len.shape[0] == n):
        raise ValueError("batch arrays differ in length")
    if n:
        end = lib().wfa_hip_batch_extent(n, _ptr(p_off), _ptr(p_len), _ptr(t_off), _ptr(t_len))   # (host threads: ~1 ms per 10 M pairs)
        if end < 0:
            raise ValueError("negative length or offset")
        if end > seqs.size:
            raise ValueError("sequence offsets run past the blob")
    return seqs, p_off, p_len, t_off, t_len, n


def _check_packed(batch):
    """A batch of 2-bit reads (datagen.to_packed2bits): byte offsets into ``packed``, lengths in bases."""
    packed = np.ascontiguousarray(batch["packed"], dtype=np.uint8)
    p_off = np.ascontiguousarray(batch["p_off"], dtype=np.int64)
    t_off = np.ascontiguousarray(batch["t_off"], dtype=np.int64)
    p_len = np.ascontiguousarray(batch["p_len"], dtype=np.int32)
    t_len = np.ascontiguousarray(batch["t_len"], dtype=np.int32)
    n = p_len.shape[0]
    if not (p_off.shape[0] == t_off.shape[0] == t_len.shape[0] == n):
        raise ValueError("batch arrays differ in length")
    if n:
        end = lib().wfa_hip_batch_extent_packed2bits(n, _ptr(p_off), _ptr(p_len), _ptr(t_off), _ptr(t_len))
        if end < 0:
            raise ValueError("negative length or offset")
        if end > packed.size:
            raise ValueError("sequence offsets run past the blob")
    return packed, p_off, p_len, t_off, t_len, n


def cigar_sprint_pretty(ops, pattern, text):
    """wfa_hip_cigar_sprint_pretty (host only): the text WFA2-lib's cigar_print_pretty prints for this op string."""
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    pattern, text = bytes(pattern), bytes(text)
    fn = lib().wfa_hip_cigar_sprint_pretty
    pb = ctypes.c_char_p(pattern) if pattern else None
    tb = ctypes.c_char_p(text) if text else None
    need = fn(_ptr(ops) if ops.size else None, ops.size, pb, len(pattern), tb, len(text), None, 0)
    if need < 0:
        raise ValueError("wfa_hip_cigar_sprint_pretty: invalid arguments")
    buf = ctypes.create_string_buffer(int(need) + 1)
    fn(_ptr(ops) if ops.size else None, ops.size, pb, len(pattern), tb, len(text), buf, need + 1)
    return buf.value.decode("ascii", "replace")


class Aligner:
    """Owns one wfa_hip_aligner_t (replaces the wavefront_aligner_t* of align.pyx:419)."""

    def __init__(self, cfg, device=0):
        L = lib()
        self._pair_caller = None
        self._pair_align = None
        self._h = L.wfa_hip_create(ctypes.byref(cfg), device)
        if not self._h:
            msg = L.wfa_hip_global_error().decode()
            rc, vmsg = validate(cfg)
            if rc == EINVAL:
                raise ValueError(vmsg)
            if rc == ENOTSUP:
                raise NotImplementedError(vmsg)
            raise NativeError(f"wfa_hip_create failed: {msg}")
        self.device = device
        self._batches = weakref.WeakSet()
        self._pair_state = None

    def upload_info(self):
        """wfa_hip_upload_info: what the upload pipeline knows about the host and what its last pipelined upload did."""
        info = np.zeros(8, np.int32)
        if lib().wfa_hip_upload_info(self._h, _ptr(info), 8) != OK:
            raise NativeError("wfa_hip_upload_info failed")
        keys = ("gpu_numa_node", "gpu_node_cpus", "numa_mode", "input_numa_node", "workers_bound", "pack_threads", "copy_threads", "process_cpus")
        return dict(zip(keys, (int(x) for x in info)))

    def close(self):
        if getattr(self, "_h", None):
            # resident batches hold device blocks of this aligner's pool: they go first (the library would also keep
            # the handle alive until the last batch is destroyed)
            for rb in list(self._batches):
                rb.close()
            lib().wfa_hip_destroy(self._h)
            self._h = None
            self._pair_caller = None
            self._pair_align = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        return lib().wfa_hip_last_error(self._h).decode()

    def _raise(self, rc, what):
        msg = self.error()
        if rc == EINVAL:
            raise ValueError(f"{what}: {msg}")
        if rc == ENOTSUP:
            raise NotImplementedError(f"{what}: {msg}")
        raise NativeError(f"{what}: {msg}")

    def set_config(self, cfg):
        rc = lib().wfa_hip_set_config(self._h, ctypes.byref(cfg))
        if rc != OK:
            self._raise(rc, "wfa_hip_set_config")

    def get_config(self):
        c = Config()
        lib().wfa_hip_get_config(self._h, ctypes.byref(c))
        return c

    def align_batch(self, batch, want_cigar, out=None):
        """Host-buffer path (wfa_hip_align_batch). Returns score, status, (ops, begin, len) or None.
        out = (score, status): int32 arrays of the batch size to write into (a loop over equally sized batches then
        touches no fresh pages: zero-filling 2 x 40 MB for 10 M pairs costs about as much as the device work)."""
        seqs, p_off, p_len, t_off, t_len, n = _check_batch(batch)
        if out is not None:
            score, status = out
            if not (score.dtype == np.int32 and status.dtype == np.int32 and score.shape == (n,) and status.shape == (n,)
                    and score.flags.c_contiguous and status.flags.c_contiguous):
                raise ValueError("out: two contiguous int32 arrays of the batch size")
        else:
            score = np.zeros(n, np.int32)
            status = np.zeros(n, np.int32)
        if want_cigar:
            cigar_off = np.zeros(n + 1, np.int64)
            np.cumsum(p_len.astype(np.int64) + t_len.astype(np.int64), out=cigar_off[1:])
            ops = np.zeros(max(int(cigar_off[-1]), 1), np.uint8)
            cbeg = np.zeros(n, np.int64)
            clen = np.zeros(n, np.int32)
        else:
            cigar_off = ops = cbeg = clen = None
        entry = lib().wfa_hip_align_batch_packed2bits if "packed" in batch else lib().wfa_hip_align_batch
        rc = entry(self._h, n, _ptr(seqs), _ptr(p_off), _ptr(p_len), _ptr(t_off),
                   _ptr(t_len), _ptr(score), _ptr(status), _ptr(ops),
                   _ptr(cigar_off), _ptr(cbeg), _ptr(clen))
        if rc != OK:
            self._raise(rc, "wfa_hip_align_batch_packed2bits" if "packed" in batch else "wfa_hip_align_batch")
        return score, status, ((ops, cbeg, clen) if want_cigar else None)

    def align_pair(self, pattern, text, want_cigar):
        """wfa_hip_align_pair: one pair of ASCII ``bytes`` per call, without NumPy arrays on the way (pywfa's loop of
        ``wavefront_align(text)``).  Returns score, status, op bytes (or None).  Through the compiled host
        (pywfa_amd/host/_host.pyx: no ctypes marshalling) when it is built, through ctypes otherwise."""
        pa = self._pair_align   # (the compiled host's bound method, looked up once per aligner: this is pywfa's one-pair-per-call loop)
        if pa is None:
            if not self._h:
                raise NativeError("wfa_hip_align_pair: the aligner is closed")
            host = compiled_host()
            if host is not None:
                self._pair_caller = host.PairCaller(self._h)
                pa = self._pair_align = self._pair_caller.align
        if pa is not None:
            rc, score, status, ops = pa(pattern, text, want_cigar)
            if rc != OK:
                self._raise(rc, "wfa_hip_align_pair")
            return score, status, ops
        st = self._pair_state
        if st is None:
            st = self._pair_state = {"score": ctypes.c_int32(0), "status": ctypes.c_int32(0), "cbeg": ctypes.c_int64(0),
                                     "clen": ctypes.c_int32(0), "ops": None, "cap": 0, "fn": lib().wfa_hip_align_pair}
            st["refs"] = (ctypes.byref(st["score"]), ctypes.byref(st["status"]), ctypes.byref(st["cbeg"]), ctypes.byref(st["clen"]))
        plen, tlen = len(pattern), len(text)
        ops = None
        if want_cigar:
            if st["cap"] < plen + tlen:
                st["cap"] = max(1024, 2 * (plen + tlen))
                st["ops"] = ctypes.create_string_buffer(st["cap"])
            ops = st["ops"]
        r = st["refs"]
        rc = st["fn"](self._h, pattern, plen, text, tlen, r[0], r[1], ops, r[2], r[3])
        if rc != OK:
            self._raise(rc, "wfa_hip_align_pair")
        if not want_cigar:
            return st["score"].value, st["status"].value, None
        b = st["cbeg"].value
        return st["score"].value, st["status"].value, ops.raw[b:b + st["clen"].value]

    def batch(self, batch):
        return ResidentBatch(self, batch)


def pack_2bit(seq, form=-1):
    """wfa_hip_pack_2bit (host only): the 2-bit words of an ASCII sequence and whether a letter outside ACGT was seen."""
    a = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else np.ascontiguousarray(seq, dtype=np.uint8)
    words = np.zeros(max((len(a) + 15) // 16, 1), np.uint32)
    rc = lib().wfa_hip_pack_2bit(_ptr(a) if len(a) else None, len(a), _ptr(words), form)
    if rc < 0:
        raise ValueError("wfa_hip_pack_2bit: invalid arguments")
    return words[:(len(a) + 15) // 16], bool(rc)


def plan_host_threads(sharers, hw_threads=None):
    """wfa_hip_plan_host_threads: (pack, copy) threads of ONE device's upload pipeline when `sharers` aligners / processes feed GPUs
    from this host (host only)."""
    pack, copy = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().wfa_hip_plan_host_threads(int(sharers), int(hw_threads or os.cpu_count() or 1), ctypes.byref(pack), ctypes.byref(copy))
    if rc != OK:
        raise ValueError("wfa_hip_plan_host_threads: invalid arguments")
    return pack.value, copy.value


def plan_shards(p_len, t_len, nshards):
    """wfa_hip_plan_shards: contiguous shards balanced by sum(p_len + t_len). Returns int64[nshards + 1] (host only)."""
    p_len = np.ascontiguousarray(p_len, dtype=np.int32)
    t_len = np.ascontiguousarray(t_len, dtype=np.int32)
    out = np.zeros(nshards + 1, np.int64)
    rc = lib().wfa_hip_plan_shards(len(p_len), _ptr(p_len), _ptr(t_len), nshards, _ptr(out))
    if rc != OK:
        raise ValueError("wfa_hip_plan_shards: invalid arguments")
    return out


class MultiAligner:
    """One aligner per device of a node (wfa_hip_multi_t): a batch is sharded over them, results land in disjoint
    slices of one set of arrays."""

    def __init__(self, cfg, devices):
        L = lib()
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        self._h = L.wfa_hip_multi_create(ctypes.byref(cfg), arr, len(self.devices))
        if not self._h:
            msg = L.wfa_hip_global_error().decode()
            rc, vmsg = validate(cfg)
            if rc == EINVAL:
                raise ValueError(vmsg)
            if rc == ENOTSUP:
                raise NotImplementedError(vmsg)
            raise NativeError(f"wfa_hip_multi_create failed: {msg}")

    def close(self):
        if getattr(self, "_h", None):
            lib().wfa_hip_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise(self, rc, what):
        msg = lib().wfa_hip_multi_last_error(self._h).decode()
        if rc == EINVAL:
            raise ValueError(f"{what}: {msg}")
        if rc == ENOTSUP:
            raise NotImplementedError(f"{what}: {msg}")
        raise NativeError(f"{what}: {msg}")

    def set_config(self, cfg):
        rc = lib().wfa_hip_multi_set_config(self._h, ctypes.byref(cfg))
        if rc != OK:
            self._raise(rc, "wfa_hip_multi_set_config")

    def align_batch(self, batch, want_cigar):
        if "packed" in batch:   # (the multi-device entry takes ASCII; 2-bit bytes would be read as letters outside ACGT)
            raise NotImplementedError("wfa_hip_multi_align_batch takes ASCII reads: 2-bit batches go through Aligner.align_batch per device")
        seqs, p_off, p_len, t_off, t_len, n = _check_batch(batch)
        score = np.zeros(n, np.int32)
        status = np.zeros(n, np.int32)
        if want_cigar:
            cigar_off = np.zeros(n + 1, np.int64)
            np.cumsum(p_len.astype(np.int64) + t_len.astype(np.int64), out=cigar_off[1:])
            ops = np.zeros(max(int(cigar_off[-1]), 1), np.uint8)
            cbeg = np.zeros(n, np.int64)
            clen = np.zeros(n, np.int32)
        else:
            cigar_off = ops = cbeg = clen = None
        rc = lib().wfa_hip_multi_align_batch(self._h, n, _ptr(seqs), _ptr(p_off), _ptr(p_len), _ptr(t_off), _ptr(t_len),
                                             _ptr(score), _ptr(status), _ptr(ops), _ptr(cigar_off), _ptr(cbeg), _ptr(clen))
        if rc != OK:
            self._raise(rc, "wfa_hip_multi_align_batch")
        return score, status, ((ops, cbeg, clen) if want_cigar else None)


class ResidentBatch:
    """A batch kept in HBM (wfa_hip_batch_t): upload + 2-bit pack once, run many times."""

    def __init__(self, aligner, batch):
        seqs, p_off, p_len, t_off, t_len, n = _check_batch(batch)
        self.aligner = aligner
        self.n = n
        self._p_len, self._t_len = p_len, t_len
        create = lib().wfa_hip_batch_create_packed2bits if "packed" in batch else lib().wfa_hip_batch_create
        self._h = create(aligner._h, n, _ptr(seqs), _ptr(p_off), _ptr(p_len), _ptr(t_off), _ptr(t_len))
        if not self._h:
            msg = aligner.error()
            if "failed:" in msg:
                raise NativeError(f"wfa_hip_batch_create: {msg}")
            raise ValueError(f"wfa_hip_batch_create: {msg}")
        aligner._batches.add(self)

    def close(self):
        if getattr(self, "_h", None):
            lib().wfa_hip_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, stream=None):
        rc = lib().wfa_hip_batch_run(self._h, stream)
        if rc != OK:
            self.aligner._raise(rc, "wfa_hip_batch_run")

    def sync(self):
        rc = lib().wfa_hip_batch_sync(self._h)
        if rc != OK:
            self.aligner._raise(rc, "wfa_hip_batch_sync")

    def results(self, want_cigar):
        n = self.n
        score = np.zeros(n, np.int32)
        status = np.zeros(n, np.int32)
        if want_cigar:
            cigar_off = np.zeros(n + 1, np.int64)
            np.cumsum(self._p_len.astype(np.int64) + self._t_len.astype(np.int64), out=cigar_off[1:])
            ops = np.zeros(max(int(cigar_off[-1]), 1), np.uint8)
            cbeg = np.zeros(n, np.int64)
            clen = np.zeros(n, np.int32)
        else:
            cigar_off = ops = cbeg = clen = None
        rc = lib().wfa_hip_batch_results(self._h, _ptr(score), _ptr(status), _ptr(ops), _ptr(cigar_off),
                                         _ptr(cbeg), _ptr(clen))
        if rc != OK:
            self.aligner._raise(rc, "wfa_hip_batch_results")
        return score, status, ((ops, cbeg, clen) if want_cigar else None)

    def last_kernel(self):
        ms = ctypes.c_float(0)
        pairs = ctypes.c_int64(0)
        rc = lib().wfa_hip_batch_last_kernel_ms(self._h, ctypes.byref(ms), ctypes.byref(pairs))
        if rc != OK:
            self.aligner._raise(rc, "wfa_hip_batch_last_kernel_ms")
        return ms.value, pairs.value

    def rle(self):
        """cigartuples + locations of the whole batch, computed on the GPU.

        Returns run_off int64[n+1], run_code uint8[total], run_len int32[total], locations int32[n,4]."""
        n = self.n
        cnt = np.zeros(n, np.int32)
        locs = np.zeros((n, 4), np.int32)
        total = lib().wfa_hip_batch_rle_counts(self._h, _ptr(cnt), _ptr(locs))
        if total < 0:
            self.aligner._raise(int(total), "wfa_hip_batch_rle_counts")
        code = np.zeros(max(int(total), 1), np.uint8)
        rlen = np.zeros(max(int(total), 1), np.int32)
        rc = lib().wfa_hip_batch_rle_runs(self._h, _ptr(code), _ptr(rlen))
        if rc != OK:
            self.aligner._raise(rc, "wfa_hip_batch_rle_runs")
        off = np.zeros(n + 1, np.int64)
        np.cumsum(cnt, out=off[1:])
        return off, code[:total], rlen[:total], locs

    def algorithmic_bytes(self):
        return int(lib().wfa_hip_batch_algorithmic_bytes(self._h))

    def fallback_pairs(self):
        return int(lib().wfa_hip_batch_fallback_pairs(self._h))
