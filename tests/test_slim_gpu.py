"""The slim form of the banded kernel (csrc/wfa_slim.hpp: the C3 hot path — gap-affine, wf-adaptive, end-to-end, reads of 1-10 kb,
score-only or piggy-back history) against the oracle and against wfa_band_kernel on the same batches (WFA_HIP_BAND_SLIM=0)."""
import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import datagen

pytestmark = pytest.mark.gpu


def ragged(seed, n, lmin, lmax, err, indel_bias=0.0):
    """n pairs with lengths drawn from [lmin, lmax]; indel_bias > 0 deletes runs from every third text so that the end diagonal
    lies far from diagonal 0 (the window has to travel, and the 128-diagonal form gets its turn)."""
    rng = np.random.default_rng(seed)
    pats, txts = [], []
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        b = datagen.generate(1, L, err, seed * 1000 + i)
        p, t = datagen.pair_strings(b, 0)
        if indel_bias and i % 3 == 0:
            for _ in range(int(indel_bias)):
                a = int(rng.integers(0, max(1, len(t) - 40)))
                t = t[:a] + t[a + int(rng.integers(5, 30)):]
        pats.append(p); txts.append(t)
    return datagen.from_strings(pats, txts)


CONFIGS = [
    dict(span="end-to-end", heuristic="adaptive"),
    dict(span="end-to-end", heuristic="adaptive", min_wavefront_length=5, max_distance_threshold=10, steps_between_cutoffs=3),
    dict(span="end-to-end", heuristic="adaptive", min_wavefront_length=30, max_distance_threshold=90, steps_between_cutoffs=2),
    dict(span="end-to-end", heuristic="adaptive", max_steps=700),
    dict(span="end-to-end", heuristic="adaptive", mismatch=4, gap_opening=4, gap_extension=2),
    dict(span="end-to-end", heuristic="adaptive", mismatch=4, gap_opening=6, gap_extension=1),
    dict(span="end-to-end", heuristic="adaptive", mismatch=3, gap_opening=4, gap_extension=1),
    dict(span="end-to-end", heuristic="adaptive", mismatch=2, gap_opening=2, gap_extension=1),
    # penalty shapes the library has no instantiation of: the slim form compiled at run time (5/8/2: six waves per SIMD; 3/6/2)
    dict(span="end-to-end", heuristic="adaptive", mismatch=5, gap_opening=6, gap_extension=2),
    dict(span="ends-free", heuristic="adaptive", mismatch=3, gap_opening=4, gap_extension=2, pattern_begin_free=10, text_end_free=20),
    # ends-free (one compare per chunk against a per-lane threshold), wavefront 0 over the free begins
    dict(span="ends-free", heuristic="adaptive", pattern_begin_free=40, pattern_end_free=30, text_begin_free=25, text_end_free=35),
    dict(span="ends-free", heuristic="adaptive", pattern_end_free=60, text_end_free=5, max_steps=900),
    # gap-affine-2p: 192 diagonals, one to three chunks active (BASELINE C4 with wf-adaptive: free pattern ends of 100)
    dict(distance="affine2p", span="ends-free", heuristic="adaptive", pattern_begin_free=100, pattern_end_free=100),
    dict(distance="affine2p", span="end-to-end", heuristic="adaptive"),
    dict(distance="affine2p", span="ends-free", heuristic="adaptive", pattern_begin_free=20, text_begin_free=30, text_end_free=50,
         min_wavefront_length=5, max_distance_threshold=20, steps_between_cutoffs=2),
    dict(distance="affine2p", span="end-to-end", heuristic="adaptive", max_steps=1500),
    # wavefronts that outgrow the first window: the 256-diagonal stage behind it (explicit history, walked in-kernel)
    dict(span="end-to-end", heuristic="adaptive", max_distance_threshold=130),
    dict(span="ends-free", heuristic="adaptive", max_distance_threshold=160, pattern_begin_free=30, text_end_free=20),
    dict(distance="affine2p", span="end-to-end", heuristic="adaptive", max_distance_threshold=110),
    # gap-affine-2p shapes the library has no instantiation of (round 6: the slim form compiled at run time — the LDS ring of the deep M
    # history takes any o2 + e2): 5/6/2/24/1 (ring of 20 rows) and 3/4/2/12/1 (ring of 10 rows, g = 1)
    dict(distance="affine2p", span="ends-free", heuristic="adaptive", mismatch=5, pattern_begin_free=50, pattern_end_free=100),
    dict(distance="affine2p", span="end-to-end", heuristic="adaptive", mismatch=3, gap_opening=4, gap_extension=2, gap_opening2=12, gap_extension2=1),
]


@pytest.mark.parametrize("scope", ["full", "score"])
@pytest.mark.parametrize("cfg_idx", range(len(CONFIGS)))
def test_slim_kernel_matches_oracle_and_band_kernel(gpu, cfg_idx, scope, monkeypatch):
    if (cfg_idx == len(CONFIGS) - 1 or (cfg_idx == len(CONFIGS) - 2 and scope == "score")) and __import__("os").environ.get("WFA_TEST_FULL") != "1":
        pytest.skip("the run-time 2p shapes: score scope with WFA_TEST_FULL=1 (the suite's time budget)")
    two = CONFIGS[cfg_idx].get("distance") == "affine2p"   # (the oracle's 2p runs are what this test takes: fewer 10 kb pairs)
    batches = [ragged(31 + cfg_idx, 96, 1100, 4000, 0.08, indel_bias=6), datagen.generate(16 if two else 48, 10000, 0.08, 4100 + cfg_idx),
               ragged(77 + cfg_idx, 64, 1200, 9000, 0.03), datagen.generate(200, 1500, 0.15, 4200 + cfg_idx)]
    whole = __import__("os").environ.get("WFA_TEST_FULL") == "1"
    for bi, batch in enumerate(batches):
        kw = common.clamp_free(dict(CONFIGS[cfg_idx], scope=scope), batch)
        oc, nc = common.configs_pair(**kw)
        full = oc.scope == 1
        o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
        monkeypatch.delenv("WFA_HIP_BAND_SLIM", raising=False)
        score, status, cigars = common.gpu_run(nc, batch, full, resident=(bi % 2 == 0))
        common.assert_same(o, score, status, cigars, batch, f"slim {kw} batch {bi}")
        if not whole and (bi + cfg_idx) % 2 == 1: continue   # (the suite's time budget: the banded kernel beside it on every other batch)
        monkeypatch.setenv("WFA_HIP_BAND_SLIM", "0")
        score0, status0, cigars0 = common.gpu_run(nc, batch, full, resident=(bi % 2 == 0))
        assert np.array_equal(score, score0) and np.array_equal(status, status0) and cigars == cigars0


@pytest.mark.parametrize("distance", ["affine", "affine2p"])
def test_slim_kernel_window_overflow_is_handed_on(gpu, distance):
    """Reads whose wavefront outgrows the window (a cut-off that keeps everything) go on to the 256-diagonal stage."""
    batch = datagen.generate(40, 6000, 0.12, 515)
    kw = dict(distance=distance, span="end-to-end", heuristic="adaptive", min_wavefront_length=10, max_distance_threshold=400, scope="full")
    oc, nc = common.configs_pair(**kw)
    o = loader.run(loader.oracle(), oc, batch)
    score, status, cigars = common.gpu_run(nc, batch, True, resident=True)
    common.assert_same(o, score, status, cigars, batch, "slim, wide wavefronts")


EXACT = [
    dict(span="end-to-end"), dict(span="ends-free", pattern_begin_free=20, pattern_end_free=10, text_begin_free=5, text_end_free=30),
    dict(span="end-to-end", max_steps=300), dict(span="end-to-end", mismatch=4, gap_opening=6, gap_extension=1), dict(span="end-to-end", mismatch=5),
    dict(distance="affine2p", span="end-to-end"), dict(distance="affine2p", span="ends-free", pattern_end_free=40, text_end_free=40),
]


@pytest.mark.parametrize("scope", ["full", "score"])
@pytest.mark.parametrize("cfg_idx", range(len(EXACT)))
def test_slim_kernel_without_a_heuristic(gpu, cfg_idx, scope, monkeypatch):
    """No heuristic: the banded stages of reads up to 1.2 kb (what the lane / segment kernels hand on, and batches they do not take)
    and the single-call path run the slim step too — unsplit launches keep the explicit history and walk it in-kernel."""
    import pywfa_amd
    batches = [ragged(500 + cfg_idx, 400, 520, 1150, 0.06, indel_bias=3), ragged(600 + cfg_idx, 300, 200, 700, 0.12),
               datagen.generate(200, 1000, 0.03, 4300 + cfg_idx)]
    for bi, batch in enumerate(batches):
        kw = common.clamp_free(dict(EXACT[cfg_idx], scope=scope), batch)
        oc, nc = common.configs_pair(**kw)
        full = oc.scope == 1
        o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
        monkeypatch.delenv("WFA_HIP_BAND_SLIM", raising=False)
        score, status, cigars = common.gpu_run(nc, batch, full, resident=(bi % 2 == 0))
        common.assert_same(o, score, status, cigars, batch, f"slim exact {kw} batch {bi}")
        monkeypatch.setenv("WFA_HIP_BAND_SLIM", "0")
        score0, status0, cigars0 = common.gpu_run(nc, batch, full, resident=(bi % 2 == 0))
        assert np.array_equal(score, score0) and np.array_equal(status, status0) and cigars == cigars0
    # one pair per call (the pinned-block path: completion flags polled by the host)
    monkeypatch.delenv("WFA_HIP_BAND_SLIM", raising=False)
    kw = common.clamp_free(dict(EXACT[cfg_idx], scope=scope), batches[1])
    a = pywfa_amd.WavefrontAligner(**{k: v for k, v in kw.items()})
    oc, nc = common.configs_pair(**kw)
    sub = datagen.subset(batches[1], np.arange(12))
    o = loader.run(loader.oracle(), oc, sub, want_cigar=(scope == "full"))
    for i in range(12):
        p, t = datagen.pair_strings(batches[1], i)
        got = a.wavefront_align(t, p)
        if a.status == 0:
            assert got == o["score"][i], (i, got, o["score"][i])
        assert a.status == o["status"][i]
        if scope == "full":
            assert bytes(a._ops) == o["cigars"][i]


@pytest.mark.parametrize("kw", [dict(span="end-to-end", heuristic="adaptive"),
                                dict(span="ends-free", heuristic="adaptive", pattern_begin_free=30, text_end_free=50),
                                dict(span="end-to-end", heuristic="adaptive", mismatch=5, gap_opening=6, gap_extension=2)])
@pytest.mark.parametrize("scope", ["full", "score"])
def test_slim_kernel_on_windows_of_long_reads(gpu, kw, scope, monkeypatch):
    """Reads whose sequences do not fit LDS (over 26 kb): the 256-diagonal slim kernel stages WINDOWS of the two sequences and moves them
    along (csrc/wfa_slim.hpp, WIN).  30-70 kb reads at 2-10 %, a pair of identical 60 kb sequences and one with a 40 kb identical stretch
    (extensions that cross several windows) against the oracle, and against wfa_band_kernel on HBM-resident sequences (WFA_HIP_BAND_NO_WIN=1)."""
    rng = np.random.default_rng(5)
    pats, txts = [], []
    for i, (L, e) in enumerate([(30000, 0.08), (45000, 0.02), (70000, 0.05), (33000, 0.10), (52000, 0.08), (28000, 0.03)]):
        b = datagen.generate(2, L, e, 900 + i)
        for j in range(2):
            p, t = datagen.pair_strings(b, j)
            pats.append(p); txts.append(t)
    same = "".join(rng.choice(list("ACGT"), size=60000))
    pats.append(same); txts.append(same)
    stretch = "".join(rng.choice(list("ACGT"), size=40000))
    b = datagen.generate(1, 8000, 0.08, 77)
    p, t = datagen.pair_strings(b, 0)
    pats.append(p[:4000] + stretch + p[4000:]); txts.append(t[:4000] + stretch + t[4000:])
    batch = datagen.from_strings(pats, txts)
    kw = common.clamp_free(dict(kw, scope=scope), batch)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
    monkeypatch.delenv("WFA_HIP_BAND_NO_WIN", raising=False)
    score, status, cigars = common.gpu_run(nc, batch, full, resident=True)
    common.assert_same(o, score, status, cigars, batch, f"slim on windows {kw}")
    monkeypatch.setenv("WFA_HIP_BAND_NO_WIN", "1")
    score0, status0, cigars0 = common.gpu_run(nc, batch, full, resident=False)
    assert np.array_equal(score, score0) and np.array_equal(status, status0) and cigars == cigars0
