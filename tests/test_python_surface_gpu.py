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
