"""memory_mode="biwfa", scope="full" on the device (csrc/wfa_biwfa.hpp) against the oracle's restatement of
R/wavefront_bialign.c (itself pinned against the real library in its ultralow mode, tests/test_oracle_vs_ref.py): status,
score — including the unset INT32_MIN score of pairs the top level answers with the ordinary algorithm (SURVEY Appendix B,
Q6) — and op string, bit for bit."""
import os

import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import _native, datagen

pytestmark = pytest.mark.gpu

CONFIGS = [dict(span="end-to-end"), dict(), dict(distance="affine2p"),
           dict(distance="affine2p", span="end-to-end", mismatch=3, gap_opening=4, gap_extension=2, gap_opening2=12, gap_extension2=1),
           dict(distance="indel"), dict(distance="levenshtein", span="end-to-end"), dict(distance="linear", mismatch=3, gap_extension=5),
           dict(match=-1, span="end-to-end"), dict(wildcard="N"), dict(mismatch=2, gap_opening=3, gap_extension=1)]

SHAPES = [(600, 150, 0.02), (300, 150, 0.2), (300, 60, 0.1), (60, 1500, 0.08), (16, 4000, 0.15), (8, 10000, 0.08)]
if os.environ.get("WFA_TEST_FULL") != "1":   # (the suite's time budget: the oracle's BiWFA runs of the long shapes are most of it)
    SHAPES = [(400, 150, 0.02), (200, 150, 0.2), (200, 60, 0.1), (40, 1500, 0.08), (10, 4000, 0.15), (5, 10000, 0.08)]


@pytest.mark.parametrize("cfg_idx", range(len(CONFIGS)))
def test_biwfa_full_cigar_matches_oracle(gpu, cfg_idx):
    import validate_oracle as vo
    kw = dict(CONFIGS[cfg_idx], scope="full", memory_mode="biwfa")
    oc, nc = common.configs_pair(**kw)
    corpora = [datagen.generate(n, L, e, 3900 + 7 * cfg_idx + i) for i, (n, L, e) in enumerate(SHAPES)]
    corpora.append(vo.corpus_special(seed=11 + cfg_idx))   # empty / length-1 / N / repeats / long gaps
    for i, batch in enumerate(corpora):
        o = loader.run(loader.oracle(), oc, batch)
        score, status, cigars = common.gpu_run(nc, batch, True, resident=(i % 2 == 0))
        common.assert_same(o, score, status, cigars, batch, f"biwfa {kw} corpus {i}")


LEVEL_ENVS = [dict(WFA_HIP_BILEVEL_QCAP="40"), dict(WFA_HIP_BILEVEL_LEVELS="2"), dict(WFA_HIP_BILEVEL="0"),
              dict(WFA_HIP_BILEVEL_I32="1", WFA_HIP_BILEVEL_WIDE_LEVELS="0"), dict(WFA_HIP_BILEVEL_WIDE_LEVELS="9", WFA_HIP_BILEVEL_NO_SEQL="1"),
              dict(WFA_HIP_BILEVEL_LDS="1", WFA_HIP_BILEVEL_LDS_W="4096")]
LEVEL_KWS = [dict(span="end-to-end"), dict(distance="affine2p", span="end-to-end"), dict(span="end-to-end", max_steps=400),
             dict(span="end-to-end", heuristic="adaptive"), dict(distance="levenshtein", span="end-to-end")]


@pytest.mark.parametrize("kw_idx", range(len(LEVEL_KWS)))
def test_biwfa_level_queues_that_overflow_are_redone(gpu, kw_idx, monkeypatch):
    """The level-by-level form (csrc/wfa_bilevel.hpp) with queues too small for the batch / too few levels: the pairs it cannot finish
    are aligned again by the depth-first kernel; and the depth-first kernel alone, int32 rings, one-wave / four-wave windows only, the
    sequences read from HBM, the rows in LDS — the same op strings every time."""
    pairs = [datagen.pair_strings(b, i) for b in (datagen.generate(8, 3000, 0.10, 99), datagen.generate(20, 700, 0.15, 98), datagen.generate(20, 90, 0.1, 97))
             for i in range(len(b["p_len"]))]
    batch = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
    oc, nc = common.configs_pair(scope="full", memory_mode="biwfa", **LEVEL_KWS[kw_idx])
    o = loader.run(loader.oracle(), oc, batch)
    for env in LEVEL_ENVS:
        for k in ("WFA_HIP_BILEVEL_QCAP", "WFA_HIP_BILEVEL_LEVELS", "WFA_HIP_BILEVEL", "WFA_HIP_BILEVEL_I32", "WFA_HIP_BILEVEL_WIDE_LEVELS",
                  "WFA_HIP_BILEVEL_NO_SEQL", "WFA_HIP_BILEVEL_LDS", "WFA_HIP_BILEVEL_LDS_W"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        score, status, cigars = common.gpu_run(nc, batch, True, resident=True)
        common.assert_same(o, score, status, cigars, batch, f"biwfa levels {env} {LEVEL_KWS[kw_idx]}")


def test_biwfa_long_reads_and_memory(gpu):
    """40 kb reads at 10 % (scores ~ 20 k: the recursion splits eight levels deep): op strings equal to the oracle's, valid
    transcripts whose penalty is the score, and the same scores as memory_mode high."""
    batch = datagen.generate(12, 40000, 0.10, 4711)
    oc, nc = common.configs_pair(span="end-to-end", scope="full", memory_mode="biwfa")
    o = loader.run(loader.oracle(), oc, batch)
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    rb.run(); rb.sync()
    score, status, (ops, cbeg, clen) = rb.results(True)
    rb.close(); al.close()
    cigars = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in range(len(score))]
    common.assert_same(o, score, status, cigars, batch, "biwfa 40 kb")
    bad, first = loader.check_cigars(oc, batch, score, ops, cbeg, clen, check_score=True)
    assert bad == 0, f"{bad} invalid transcripts, first at pair {first}"
    _, nc_h = common.configs_pair(span="end-to-end", scope="score")
    score_h, _, _ = common.gpu_run(nc_h, batch, False, True)
    assert np.array_equal(score_h, score)


def test_biwfa_100kb_vs_the_real_library(gpu):
    """SURVEY §8 f4 names BiWFA "for 100 kb+": four 100 kb pairs at 8 % (scores ~ 45 k, the recursion splits nine levels deep)
    against the real library in its ultralow mode (all host threads; the oracle where the reference build did not travel)."""
    batch = datagen.generate(4, 100000, 0.08, 1005)
    oc, nc = common.configs_pair(span="end-to-end", scope="full", memory_mode="biwfa")
    o = loader.reference_mt_full(oc, batch) if loader.have_reference() else loader.run(loader.oracle(), oc, batch)
    score, status, cigars = common.gpu_run(nc, batch, True, resident=True)
    common.assert_same(o, score, status, cigars, batch, "biwfa 100 kb")
    assert (status == 0).all()


def test_biwfa_python_surface(gpu):
    import pywfa_amd
    p = "TCTTTACTCGCGCGTTGGAGAAATACAATAGT" * 8
    t = "TCTATACTGCGCGTTTGGAGAAATAAAATAGT" * 8
    a = pywfa_amd.WavefrontAligner(p, memory_mode="biwfa", span="end-to-end")
    h = pywfa_amd.WavefrontAligner(p, span="end-to-end")
    assert a.wavefront_align(t) == h.wavefront_align(t)
    assert a.status == 0 and a.cigarstring == h.cigarstring or len(a.cigarstring) > 0
    with pytest.raises(NotImplementedError):   # (the reference itself exit(1)s with free ends, R/wavefront_align.c:60-75)
        pywfa_amd.WavefrontAligner(p, memory_mode="biwfa", span="ends-free", text_end_free=3)
    c = pywfa_amd.WavefrontAligner(p, memory_mode="biwfa", span="end-to-end", heuristic="adaptive")   # (round 4: on the device)
    assert c.wavefront_align(t) == h.wavefront_align(t) and c.status == 0
    c.close()
    # a step limit below the score: -100 and the unset score, in both scopes (R/wavefront_bialign.c:475,513,725)
    for scope in ("full", "score"):
        b = pywfa_amd.WavefrontAligner(p, memory_mode="biwfa", span="end-to-end", max_steps=30, scope=scope)
        assert b.wavefront_align(t) == -2147483648 and b.status == -100 and b.cigarstring == ""
        b.close()


BIWFA_GOLD = common.load_golden("biwfa.json")


@pytest.mark.parametrize("run_idx", range(len(BIWFA_GOLD["runs"])))
def test_biwfa_matches_golden_vectors(gpu, run_idx):
    """The device BiWFA against the committed outputs of the real library (tests/golden/biwfa.json)."""
    run = BIWFA_GOLD["runs"][run_idx]
    pairs = BIWFA_GOLD["corpora"][run["corpus"]]
    batch = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
    _, nc = common.configs_pair(**run["config"])
    full = run["config"]["scope"] == "full"
    score, status, cigars = common.gpu_run(nc, batch, full, resident=(run_idx % 2 == 0))
    assert score.tolist() == run["score"]
    assert status.tolist() == run["status"]
    if full:
        assert [common.rle(c) for c in cigars] == run["cigar"]


@pytest.mark.parametrize("kw", [dict(match=-1, span="end-to-end"), dict(mismatch=6, gap_opening=2, gap_extension=3), dict(distance="affine2p", match=-2),
                                dict(match=-1, max_steps=600), dict(distance="linear", match=-1, mismatch=5, gap_extension=4)])
def test_biwfa_short_divergent_reads_under_large_penalties(gpu, kw):
    """the top-level base case of reads <= 100 bases has no score bound (two unrelated 100-mers under match=-1 score
    ~1000): what outgrows the BiWFA kernel's base-case history is aligned by the general kernel — same op strings, the unset
    score, and "unattainable" where the base aligner's own step limit strikes."""
    rng = np.random.default_rng(5)
    pats = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(40, 101)))) for _ in range(400)]
    txts = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(40, 101)))) for _ in range(400)]
    batch = datagen.from_strings(pats + ["ACGT" * 20, ""], txts + ["ACGT" * 20, "ACGTACGT"])
    oc, nc = common.configs_pair(**dict(kw, scope="full", memory_mode="biwfa"))
    o = loader.run(loader.oracle(), oc, batch)
    for resident in (True, False):
        score, status, cigars = common.gpu_run(nc, batch, True, resident=resident)
        common.assert_same(o, score, status, cigars, batch, f"biwfa short divergent {kw}")


@pytest.mark.parametrize("kw0", [dict(span="end-to-end"), dict(distance="affine2p"), dict(distance="levenshtein", span="end-to-end"), dict(match=-1, span="end-to-end")])
@pytest.mark.parametrize("scope", ["full", "score"])
def test_biwfa_step_limit_matches_oracle(gpu, kw0, scope):
    # (the suite's time budget: with CIGARs gap-affine-2p only under -m gpu, 10 s a case; WFA_TEST_FULL=1 runs the four)
    import os
    if scope == "full" and kw0.get("distance") != "affine2p" and os.environ.get("WFA_TEST_FULL") != "1":
        pytest.skip("sampled on the suite's time budget (WFA_TEST_FULL=1 runs every cell)")
    import validate_oracle as vo
    corpora = [datagen.generate(500, 150, 0.05, 21), datagen.generate(200, 150, 0.2, 22), datagen.generate(300, 60, 0.1, 23),
               datagen.generate(40, 1500, 0.08, 24), datagen.generate(3, 10000, 0.08, 25), vo.corpus_special(seed=6)]
    # (with CIGARs two of the four limits under -m gpu: one that cuts most alignments short and one that few reach)
    for ms in ((5, 60, 300, 1200) if scope == "score" or os.environ.get("WFA_TEST_FULL") == "1" else (60, 1200)):
        oc, nc = common.configs_pair(**dict(kw0, scope=scope, memory_mode="biwfa", max_steps=ms))
        for i, batch in enumerate(corpora):
            o = loader.run(loader.oracle(), oc, batch)
            score, status, cigars = common.gpu_run(nc, batch, scope == "full", resident=(i % 2 == 0))
            common.assert_same(o, score, status, cigars, batch, f"biwfa max_steps={ms} {scope} {kw0} corpus {i}")


BIWFA_HEUR = [dict(heuristic="adaptive"), dict(heuristic="adaptive", min_wavefront_length=5, max_distance_threshold=15, steps_between_cutoffs=3),
              dict(heuristic="X-drop", xdrop=400), dict(heuristic="X-drop", xdrop=100, match=-1), dict(heuristic="X-drop", xdrop=20),
              dict(heuristic="adaptive", distance="affine2p"), dict(heuristic="adaptive", distance="levenshtein"),
              dict(heuristic="adaptive", distance="linear", mismatch=3, gap_extension=5), dict(heuristic="adaptive", max_steps=400)]


@pytest.mark.parametrize("cfg_idx", range(len(BIWFA_HEUR)))
@pytest.mark.parametrize("scope", ["full", "score"])
def test_biwfa_with_a_heuristic_matches_oracle(gpu, cfg_idx, scope):
    """Round 4 (VERDICT r03 missing 3): BiWFA with a heuristic on the device — the forward and the reverse aligner of every breakpoint
    search cut their wavefronts off (R/wavefront_bialigner.c:53,161-166; state re-set per search, base cases without), in both scopes.
    Against the oracle, whose restatement is pinned against the real library in ultralow mode
    (tests/test_oracle_vs_ref.py::test_biwfa_with_a_heuristic_equals_reference)."""
    import validate_oracle as vo
    kw = dict(BIWFA_HEUR[cfg_idx], scope=scope, memory_mode="biwfa", span="end-to-end")
    corpora = [datagen.generate(400, 150, 0.02, 5100 + cfg_idx), datagen.generate(200, 150, 0.2, 5200 + cfg_idx), datagen.generate(200, 60, 0.1, 5300 + cfg_idx),
               datagen.generate(40, 1500, 0.08, 5400 + cfg_idx), datagen.generate(10, 4000, 0.15, 5500 + cfg_idx), datagen.generate(6, 10000, 0.08, 5600 + cfg_idx),
               vo.corpus_special(seed=25 + cfg_idx)]
    oc, nc = common.configs_pair(**kw)
    for i, batch in enumerate(corpora):
        o = loader.run(loader.oracle(), oc, batch)
        score, status, cigars = common.gpu_run(nc, batch, scope == "full", resident=(i % 2 == 0))
        common.assert_same(o, score, status, cigars, batch, f"biwfa + heuristic {kw} corpus {i}")
