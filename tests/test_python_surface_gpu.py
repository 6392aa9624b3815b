"""pywfa_amd.WavefrontAligner replays the recorded behaviour of pywfa.WavefrontAligner (the
reference's own tests/test.py cases, README examples, FASTA fixtures, property surface, exceptions)."""
import pytest

import common
import golden_runner
import pywfa_amd

pytestmark = pytest.mark.gpu

SURFACE = common.load_golden("python_surface.json")

# documented deviations: none of the recorded cases hit them
SKIP = set()


@pytest.mark.parametrize("idx", range(len(SURFACE)))
def test_surface_case(gpu, idx):
    entry = SURFACE[idx]
    case = entry["case"]
    if case["name"] in SKIP:
        pytest.skip("documented deviation")
    got = golden_runner.run_case(pywfa_amd.WavefrontAligner, case)
    exp = entry["expected"]
    assert len(got) == len(exp), case["name"]
    for i, (g, e) in enumerate(zip(got, exp)):
        assert g == e, f"{case['name']} step {i}: expected {str(e)[:400]} got {str(g)[:400]}"


def test_batch_api_matches_single(gpu):
    a = pywfa_amd.WavefrontAligner("TCTTTACTCGCGCGTTGGAGAAATACAATAGT")
    texts = ["TCTATACTGCGCGTTTGGAGAAATAAAATAGT", "TCTTTACTCGCGCGTTGGAGAAATACAATAGT", "tctttactcgcgcgttggag"]
    out = a.wavefront_align_batch(texts)
    for t, s, c in zip(texts, out["score"], out["cigarstrings"]):
        assert a.wavefront_align(t) == s
        assert a.cigarstring == c
    assert out["cigarstrings"][0] == "3M1X4M1D7M1I9M1X6M" and out["score"][0] == -24


def test_setters_revalidate(gpu):
    a = pywfa_amd.WavefrontAligner("ACGTACGT")
    a.scope = "score"
    assert a.scope == "score" and a("ACGTTCGT").cigartuples == []
    a.scope = "full"
    a.span = "end-to-end"
    a.mismatch_penalty = 2
    assert a.wavefront_align("ACGTTCGT") == -2 and a.cigarstring == "4M1X3M"
    with pytest.raises(ValueError):
        a.scope = "half"
    with pytest.raises(ValueError):
        a.mismatch_penalty = 0        # the reference exit(1)s here
    assert a.mismatch_penalty == 2
    a.memory_mode = "med"
    assert a.memory_mode == "medium"
    a.max_steps = 1
    assert a.wavefront_align("TTTTTTTT") == -1 and a.status == -100


def test_device_side_cigartuples_and_locations(gpu):
    """the run-length encoded cigartuples and the locations of a whole batch computed on
    the GPU equal what the per-pair Python code of the reference's class derives from the op string."""
    import numpy as np
    import validate_oracle as vo
    from pywfa_amd import datagen
    from pywfa_amd.align import _ops_to_tuples, _flank_scan
    batches = [datagen.generate(3000, 150, 0.05, 41), datagen.generate(40, 3000, 0.08, 42), vo.corpus_special(seed=5)]
    for kw in (dict(), dict(span="end-to-end"), dict(distance="affine2p", pattern_end_free=0, text_end_free=0)):
        a = pywfa_amd.WavefrontAligner(**kw)
        for batch in batches:
            res = a.align_batch_results(batch)
            out = a.align_batch(batch)
            assert np.array_equal(res.score, out["score"]) and np.array_equal(res.status, out["status"])
            n = len(res)
            for i in list(range(min(n, 400))) + list(range(max(0, n - 50), n)):
                ct = _ops_to_tuples(out["cigar_ops"][i])
                assert res.cigartuples(i) == ct, i
                pl, tl = int(batch["p_len"][i]), int(batch["t_len"][i])
                if not ct or pl == 0 or tl == 0:
                    exp = (0, 0, 0, 0)
                else:
                    _, _, ps, pe, ts, te = _flank_scan(ct, 1, 1, tl, pl)
                    exp = (ps, pe, ts, te)
                assert tuple(int(x) for x in res.locations[i]) == exp, (i, ct[:6])
        r0 = a.align_batch_results(batches[0])[0]
        single = a(r0.text, r0.pattern)
        assert (r0.cigartuples, r0.score, r0.text_start, r0.text_end, r0.pattern_start, r0.pattern_end) == \
               (single.cigartuples, single.score, single.text_start, single.text_end, single.pattern_start, single.pattern_end)
