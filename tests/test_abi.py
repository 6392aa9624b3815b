"""The C-ABI library loads and exports every symbol include/wfa_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from pywfa_amd import _native

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "wfa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wfa_hip_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    syms = header_symbols()
    assert len(syms) >= 19
    lib = ctypes.CDLL(_native.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"libwfa_hip.so does not export {s}"
    assert sorted(_native.SYMBOLS) == syms, "pywfa_amd/_native.py binds a different symbol set than the header declares"


def test_abi_version_and_default_config():
    L = _native.lib()
    assert L.wfa_hip_abi_version() == _native.ABI_VERSION
    c = _native.default_config()
    # pywfa defaults (align.pyx:309-334)
    assert (c.distance, c.match, c.mismatch, c.gap_opening, c.gap_extension) == (3, 0, 4, 6, 2)
    assert (c.gap_opening2, c.gap_extension2) == (24, 1)
    assert (c.scope, c.span, c.heuristic, c.memory_mode, c.max_steps, c.wildcard) == (1, 1, 0, 0, 0, -1)
    assert (c.min_wavefront_length, c.max_distance_threshold, c.steps_between_cutoffs, c.xdrop) == (10, 50, 1, 20)
    assert ctypes.sizeof(_native.Config) == 22 * 4


@pytest.mark.parametrize("field,value,code", [
    ("match", 1, _native.EINVAL), ("mismatch", 0, _native.EINVAL), ("gap_opening", -1, _native.EINVAL),
    ("gap_extension", 0, _native.EINVAL), ("scope", 7, _native.EINVAL), ("span", 3, _native.EINVAL),
    ("heuristic", 9, _native.EINVAL), ("distance", 7, _native.EINVAL),
    ("memory_mode", 9, _native.EINVAL), ("pattern_begin_free", -2, _native.EINVAL), ("wildcard", 300, _native.EINVAL),
])
def test_validate_rejects(field, value, code):
    """Invalid penalties return an error code where the reference exit(1)s (wavefront_penalties.c:101-112)."""
    c = _native.default_config()
    setattr(c, field, value)
    rc, msg = _native.validate(c)
    assert rc == code and msg


def test_validate_biwfa():
    """memory_mode biwfa (full CIGAR, step limit and — round 4 — heuristics included) is on the accelerated path without free ends;
    those are refused with ENOTSUP (the reference itself exit(1)s on free ends, wavefront_align.c:60-75)."""
    c = _native.default_config()
    c.memory_mode = 3
    assert _native.validate(c)[0] == _native.OK
    c.max_steps = 50
    assert _native.validate(c)[0] == _native.OK
    c.scope = 0
    assert _native.validate(c)[0] == _native.OK
    for heur in (1, 2):
        c = _native.default_config()
        c.memory_mode = 3
        c.heuristic = heur
        assert _native.validate(c)[0] == _native.OK
    c = _native.default_config()
    c.memory_mode = 3
    c.text_end_free = 4
    assert _native.validate(c)[0] == _native.ENOTSUP


def test_validate_single_component_metrics():
    for d in (0, 1, 2):
        c = _native.default_config()
        c.distance = d
        assert _native.validate(c)[0] == _native.OK
    c = _native.default_config()
    c.distance, c.heuristic = 1, 2   # X-drop with edit: the reference exit(1)s (wavefront_align.c:80-85)
    assert _native.validate(c)[0] == _native.EINVAL


def test_validate_accepts_affine2p():
    c = _native.default_config()
    c.distance = 4
    assert _native.validate(c)[0] == _native.OK
    c.gap_extension2 = 0
    assert _native.validate(c)[0] == _native.EINVAL


def test_no_gpu_fails_loudly():
    """Without a HIP device the product path raises (there is no CPU fallback)."""
    if _native.lib().wfa_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    import pywfa_amd
    with pytest.raises(_native.NativeError):
        pywfa_amd.WavefrontAligner("ACGT")


def test_product_does_not_use_oracle():
    """Nothing under pywfa_amd/ may import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "pywfa_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".c", ".sh")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in txt.lower(), (dirpath, fn)
                assert "libwfa_ref" not in txt, (dirpath, fn)


def test_shard_planner_properties():
    """wfa_hip_plan_shards (host only): contiguous shards that cover the batch, balanced by bases, for any shard count."""
    import numpy as np
    from pywfa_amd import _native
    rng = np.random.default_rng(1)
    for n in (0, 1, 5, 1000, 20000):
        pl = rng.integers(0, 3000, n).astype(np.int32)
        tl = rng.integers(0, 3000, n).astype(np.int32)
        for k in (1, 2, 3, 8, 13):
            sb = _native.plan_shards(pl, tl, k)
            assert sb[0] == 0 and sb[-1] == n and (np.diff(sb) >= 0).all() and len(sb) == k + 1
            if n >= 1000:
                w = pl.astype(np.int64) + tl + 16
                loads = np.array([w[sb[i]:sb[i + 1]].sum() for i in range(k)])
                assert loads.max() - loads.min() <= 2 * w.max() + 1
    import pytest
    with pytest.raises(ValueError):
        _native.plan_shards(np.zeros(3, np.int32), np.zeros(3, np.int32), 0)


def test_host_pack_2bit_all_forms():
    """wfa_hip_pack_2bit (host only): every form (plain C / AVX2 / AVX-512BW, a form the CPU lacks falls back) gives the
    words of the device layout — base j of word w in bits 2 j, code (c >> 1) & 3, zero beyond the end — and reports
    letters outside ACGT; every length around the 16 / 32 / 64-base rounds, reads that end at the end of their buffer."""
    import numpy as np
    from pywfa_amd import _native

    def ref(s):
        a = np.frombuffer(s, dtype=np.uint8)
        n = len(a); nw = (n + 15) // 16
        c = np.zeros(nw * 16, np.uint64); c[:n] = (a >> 1) & 3
        w = (c.reshape(nw, 16) << (2 * np.arange(16, dtype=np.uint64))).sum(1).astype(np.uint32) if nw else np.zeros(0, np.uint32)
        return w, any(ch not in b"ACGT" for ch in s)

    rng = np.random.default_rng(5)
    for form in (-1, 0, 1, 2):
        for L in list(range(0, 200)) + [255, 256, 257, 1000, 10007]:
            s = bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8))
            if L and rng.random() < 0.3:
                i = int(rng.integers(L)); s = s[:i] + bytes([int(rng.choice(list(b"NacgtRY\x00\xff")))]) + s[i + 1:]
            w, bad = _native.pack_2bit(s, form)
            rw, rbad = ref(s)
            assert bad == rbad and np.array_equal(w, rw), (form, L, s)


def _expand_cigar(cigarstring):
    ops = bytearray()
    for n, c in re.findall(r"(\d+)([MXID])", cigarstring):
        ops += c.encode() * int(n)
    return bytes(ops)


def test_cigar_sprint_pretty_matches_the_reference_text():
    """wfa_hip_cigar_sprint_pretty (host only) against every cigar_print_pretty text recorded from the reference
    (tests/golden/python_surface.json, written by tools/make_golden.py)."""
    import json
    import numpy as np
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "python_surface.json")))
    seen = 0
    for rec in cases:
        pattern = (rec["case"]["ctor"].get("pattern") or "").upper()
        text, cigar = None, None
        for st, out in zip(rec["case"]["steps"], rec["expected"]):
            if st["op"] == "align" and "aligner" in out:
                text, cigar = st["text"], out["aligner"]["cigarstring"]
                if st.get("pattern") is not None:
                    pattern = st["pattern"].upper()
            elif st["op"] == "pretty_print" and "text" in out and text is not None:
                ops = np.frombuffer(_expand_cigar(cigar), dtype=np.uint8)
                got = _native.cigar_sprint_pretty(ops, pattern.encode(), text.encode())
                assert got == out["text"], rec["case"]["name"]
                seen += 1
    assert seen >= 1
    # truncation like snprintf, and the unaligned tails marked '?'
    L = _native.lib()
    buf = ctypes.create_string_buffer(8)
    need = L.wfa_hip_cigar_sprint_pretty(None, 0, ctypes.c_char_p(b"AC"), 2, ctypes.c_char_p(b"ACG"), 3, buf, 8)
    assert need > 8 and buf.value == b"      A"
    assert _native.cigar_sprint_pretty(np.zeros(0, np.uint8), b"AC", b"ACG").endswith(
        "      PATTERN    AC\n                 ???\n      TEXT       ACG\n")


def test_packed2bits_helper_layout():
    """datagen.to_packed2bits writes the reference's packed form (wavefront_sequences.c:102-139): four bases per byte,
    base j in bits 2j..2j+1, A 0 / C 1 / G 2 / T 3; the C helper and the NumPy form agree."""
    import numpy as np
    from pywfa_amd import datagen
    b = datagen.from_strings(["ACGTA", "", "TTTTGGGGC"], ["GATTACA", "C", ""])
    pk = datagen.to_packed2bits(b)
    lut = "ACGT"
    for i in range(3):
        for off, ln, src in ((pk["p_off"][i], pk["p_len"][i], b["p_off"][i]), (pk["t_off"][i], pk["t_len"][i], b["t_off"][i])):
            dec = "".join(lut[(pk["packed"][off + j // 4] >> (2 * (j % 4))) & 3] for j in range(ln))
            assert dec == b["seqs"][src:src + ln].tobytes().decode()
    saved = datagen._synth
    try:
        datagen._synth = False
        pk2 = datagen.to_packed2bits(b)
    finally:
        datagen._synth = saved
    assert np.array_equal(pk2["packed"], pk["packed"]) and np.array_equal(pk2["t_off"], pk["t_off"])
    with pytest.raises(ValueError):
        datagen.to_packed2bits(datagen.from_strings(["ACGN"], ["ACGT"]))


def test_host_thread_plan_fits_the_host_for_eight_devices():
    """each device's upload pipeline takes its share of the host — 8 devices on 256 logical CPUs must not ask
    for more threads than there are CPUs (round 4: 8 x (32 + 8) = 320), one device alone keeps 32 + 8 (host only, no GPU)."""
    import ctypes
    L = _native.lib()
    L.wfa_hip_plan_host_threads.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    pack, copy = ctypes.c_int(0), ctypes.c_int(0)
    for hw in (8, 64, 96, 128, 256, 384):
        for sharers in (1, 2, 4, 8):
            assert L.wfa_hip_plan_host_threads(sharers, hw, ctypes.byref(pack), ctypes.byref(copy)) == 0
            assert pack.value >= 1 and copy.value >= 1
            if hw >= 4 * sharers:
                assert sharers * (pack.value + copy.value) <= hw, (hw, sharers, pack.value, copy.value)
    assert L.wfa_hip_plan_host_threads(1, 256, ctypes.byref(pack), ctypes.byref(copy)) == 0 and (pack.value, copy.value) == (32, 8)
    assert L.wfa_hip_plan_host_threads(8, 256, ctypes.byref(pack), ctypes.byref(copy)) == 0 and (pack.value, copy.value) == (16, 8)
    assert L.wfa_hip_plan_host_threads(0, 256, ctypes.byref(pack), ctypes.byref(copy)) != 0
