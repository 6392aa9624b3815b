"""memory_mode="biwfa", scope="full" on the device (csrc/wfa_biwfa.hpp) against the oracle's restatement of
R/wavefront_bialign.c (itself pinned against the real library in its ultralow mode, tests/test_oracle_vs_ref.py): status,
score — including the unset INT32_MIN score of pairs the top level answers with the ordinary algorithm (SURVEY Appendix B,
Q6) — and op string, bit for bit."""
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


def test_biwfa_python_surface(gpu):
    import pywfa_amd
    p = "TCTTTACTCGCGCGTTGGAGAAATACAATAGT" * 8
    t = "TCTATACTGCGCGTTTGGAGAAATAAAATAGT" * 8
    a = pywfa_amd.WavefrontAligner(p, memory_mode="biwfa", span="end-to-end")
    h = pywfa_amd.WavefrontAligner(p, span="end-to-end")
    assert a.wavefront_align(t) == h.wavefront_align(t)
    assert a.status == 0 and a.cigarstring == h.cigarstring or len(a.cigarstring) > 0
    for kw in (dict(heuristic="adaptive"), dict(max_steps=100), dict(span="ends-free", text_end_free=3)):
        with pytest.raises(NotImplementedError):
            pywfa_amd.WavefrontAligner(p, memory_mode="biwfa", **kw)


BIWFA_GOLD = common.load_golden("biwfa.json")


@pytest.mark.parametrize("run_idx", range(len(BIWFA_GOLD["runs"])))
def test_biwfa_matches_golden_vectors(gpu, run_idx):
    """The device BiWFA against the committed outputs of the real library (tests/golden/biwfa.json)."""
    run = BIWFA_GOLD["runs"][run_idx]
    pairs = BIWFA_GOLD["corpora"][run["corpus"]]
    batch = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
    _, nc = common.configs_pair(**run["config"])
    score, status, cigars = common.gpu_run(nc, batch, True, resident=(run_idx % 2 == 0))
    assert score.tolist() == run["score"]
    assert status.tolist() == run["status"]
    assert [common.rle(c) for c in cigars] == run["cigar"]
