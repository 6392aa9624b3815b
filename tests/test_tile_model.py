"""The blocked schedule of the tiled wide-wavefront kernel (csrc/wfa_tile.hpp) on the CPU: tools/tile_model.cpp runs the same
ring-slot / range / tile-load / write-back rules and the same compute-next (shared header csrc/wfa_tile_cell.hpp) one pair at a
time; here it is compared with the oracle.  What this pins down is the exactness argument of the schedule (supersets of the
reference's ranges, no per-step trimming, the taint check that hands a pair on when trimming would have changed a value) for
several geometries (T steps per super-step, tile width) and penalty sets — before any GPU is involved."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest

from oracle import loader
from pywfa_amd import datagen

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    out = tmp_path_factory.mktemp("tile_model") / "libtile_model.so"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "pywfa_amd", "csrc"),
                    os.path.join(ROOT, "tools", "tile_model.cpp"), "-o", str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    lib.tile_model_align.restype = ctypes.c_int
    return lib


def run_model(lib, kw, batch, T, Wt, force_careful=0):
    """-> list of (status, score, ops bytes or None, careful passes) per pair; status as the kernel's end_reason."""
    two = kw.get("distance", "affine") == "affine2p"
    x, o, e = kw.get("mismatch", 4), kw.get("gap_opening", 6), kw.get("gap_extension", 2)
    o2, e2 = kw.get("gap_opening2", 24), kw.get("gap_extension2", 1)
    g = math.gcd(math.gcd(x, o + e), e)
    if two:
        g = math.gcd(g, math.gcd(o2 + e2, e2))
    X, OE, E = x // g, (o + e) // g, e // g
    OE2, E2 = ((o2 + e2) // g, e2 // g) if two else (0, 0)
    full = kw.get("scope", "full") == "full"
    ef = kw.get("span", "ends-free") == "ends-free"
    ms = kw.get("max_steps", 0)
    ms_t = 2**31 - 1 if ms <= 0 else max(1, -(-ms // g))
    res = []
    seqs = np.ascontiguousarray(batch["seqs"], dtype=np.uint8)
    for i in range(len(batch["p_len"])):
        pl, tl = int(batch["p_len"][i]), int(batch["t_len"][i])
        P = seqs[int(batch["p_off"][i]):int(batch["p_off"][i]) + pl].copy()
        Tx = seqs[int(batch["t_off"][i]):int(batch["t_off"][i]) + tl].copy()
        ops = np.zeros(pl + tl + 8, np.uint8)
        end_t, end_k, nops, cp = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        st = lib.tile_model_align(X, OE, E, OE2, E2, T, Wt, int(full), int(ef), kw.get("pattern_begin_free", 0), kw.get("pattern_end_free", 0),
                                  kw.get("text_begin_free", 0), kw.get("text_end_free", 0), ms_t,
                                  P.ctypes.data_as(ctypes.c_void_p), pl, Tx.ctypes.data_as(ctypes.c_void_p), tl,
                                  ctypes.byref(end_t), ctypes.byref(end_k), ops.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nops),
                                  ctypes.byref(cp), force_careful)
        assert st >= 0, f"model self-check failed ({st}) on pair {i}"
        if st == 1:
            res.append((0, -end_t.value * g, ops[:nops.value].tobytes() if full else None, cp.value))
        elif st == 4:
            res.append((-100, -ms, b"" if full else None, cp.value))
        else:
            res.append((None, None, None, cp.value))   # handed on
    return res


def compare(lib, kw, batch, T, Wt, max_handed=0.0, force_careful=0):
    o = loader.run(loader.oracle(), loader.make_config(**kw), batch, want_cigar=kw.get("scope", "full") == "full")
    res = run_model(lib, kw, batch, T, Wt, force_careful)
    handed = 0
    for i, (st, sc, ops, _) in enumerate(res):
        if st is None:
            handed += 1
            continue
        assert (st, sc) == (int(o["status"][i]), int(o["score"][i])), (kw, T, Wt, i, st, sc, int(o["status"][i]), int(o["score"][i]))
        if ops is not None and o.get("cigars") is not None:
            assert ops == o["cigars"][i], (kw, T, Wt, i)
    assert handed <= max_handed * len(res), (kw, T, Wt, handed, len(res))
    return handed, sum(r[3] for r in res)


GEOMS = [(2, 64), (4, 64), (8, 64), (8, 128), (16, 128), (30, 128)]

CONFIGS = [
    dict(span="end-to-end", scope="full"),
    dict(span="end-to-end", scope="score", mismatch=2, gap_opening=3, gap_extension=1),
    dict(span="end-to-end", scope="full", mismatch=5, gap_opening=0, gap_extension=3),
    dict(span="end-to-end", scope="full", mismatch=6, gap_opening=5, gap_extension=3),
    dict(span="end-to-end", scope="full", mismatch=1, gap_opening=1, gap_extension=1),
    dict(span="ends-free", scope="full", pattern_begin_free=30, pattern_end_free=40, text_begin_free=20, text_end_free=10),
    dict(span="ends-free", scope="full", pattern_begin_free=0, pattern_end_free=0, text_begin_free=50, text_end_free=50),
    dict(distance="affine2p", span="end-to-end", scope="full"),
    dict(distance="affine2p", span="ends-free", scope="full", pattern_begin_free=30, pattern_end_free=40, text_begin_free=20, text_end_free=10),
    dict(distance="affine2p", span="end-to-end", scope="full", mismatch=3, gap_opening=4, gap_extension=2, gap_opening2=12, gap_extension2=1),
    dict(distance="affine2p", span="end-to-end", scope="full", mismatch=2, gap_opening=2, gap_extension=2, gap_opening2=10, gap_extension2=1),
    dict(span="end-to-end", scope="full", max_steps=60),
    dict(distance="affine2p", span="end-to-end", scope="score", max_steps=45),
]


def clamp(kw, batch):
    pl, tl = int(batch["p_len"].min()), int(batch["t_len"].min())
    kw = dict(kw)
    for k, lim in (("pattern_begin_free", pl), ("pattern_end_free", pl), ("text_begin_free", tl), ("text_end_free", tl)):
        if kw.get(k, 0) > lim:
            kw[k] = lim
    return kw


@pytest.mark.parametrize("ci", range(len(CONFIGS)))
def test_blocked_schedule_equals_oracle(model, ci):
    """Reads of ~400 bases at 10 %: the schedule may hand on a pair only when the trimming check fires (rare); every other pair is
    bit-exact in score, status and op string, for every geometry."""
    batch = datagen.generate(24, 400, 0.10, 500 + ci)
    kw = clamp(CONFIGS[ci], batch)
    for T, Wt in GEOMS:
        compare(model, kw, batch, T, Wt, max_handed=0.15)


def ragged(n, seed):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    pats, txts = [], []
    for i in range(n):
        pl = int(rng.integers(0, 70))
        p = alpha[rng.integers(0, 4, pl)]
        mode = i % 4
        if mode == 0:
            t = alpha[rng.integers(0, 4, int(rng.integers(0, 70)))]      # unrelated
        elif mode == 1:
            t = p[:int(rng.integers(0, pl + 1))]                          # a prefix
        elif mode == 2:
            t = np.concatenate([alpha[rng.integers(0, 4, int(rng.integers(0, 20)))], p])   # an overhang
        else:
            t = p.copy()
            for _ in range(int(rng.integers(0, 8))):
                if len(t):
                    t[int(rng.integers(0, len(t)))] = alpha[int(rng.integers(0, 4))]
        pats.append(p.tobytes().decode()); txts.append(t.tobytes().decode())
    return datagen.from_strings(pats, txts)


@pytest.mark.parametrize("ci", [0, 1, 2, 4, 7, 9])
def test_short_and_unrelated_sequences(model, ci):
    """Sequences of 0-70 bases, unrelated / prefixes / overhangs: wavefronts run into the ends of the diagonals range and cells past
    the sequence ends are common — where trimming can matter.  Whatever the schedule does not hand on must be exact; and with the
    trimming statistics collected in every super-step (force_careful) the outcome is the same."""
    batch = ragged(60, 900 + ci)
    kw = dict(CONFIGS[ci])
    for T, Wt in GEOMS[:4]:
        h0, _ = compare(model, kw, batch, T, Wt, max_handed=1.0)
        h1, _ = compare(model, kw, batch, T, Wt, max_handed=1.0, force_careful=1)
        assert h0 == h1


def test_trimming_check_fires_rarely_on_long_reads(model):
    """2.5 kb at 10 %, as tests/test_wide_gpu.py uses: cells past the end appear (the careful pass runs) but between in-bounds cells:
    nothing is handed on."""
    batch = datagen.generate(4, 2500, 0.10, 77)
    handed, careful = compare(model, dict(span="end-to-end", scope="full"), batch, 8, 128)
    assert handed == 0
    handed, careful2 = compare(model, dict(distance="affine2p", span="end-to-end", scope="score"), batch, 8, 128)
    assert handed == 0
