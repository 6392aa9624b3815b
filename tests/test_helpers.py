"""Python-side result helpers against vectors recorded from the reference (align.pyx:17-295)."""
import common
import pywfa_amd
from pywfa_amd.align import AlignmentResult

H = common.load_golden("helpers.json")


def _t(ct):
    return [tuple(x) for x in ct]


def test_cigartuples_to_str():
    for v in H["cigartuples_to_str"]:
        assert pywfa_amd.cigartuples_to_str(_t(v["in"])) == v["out"]
    assert pywfa_amd.cigartuples_to_str([]) == ""
    assert pywfa_amd.cigartuples_to_str(None) == ""


def test_elide_mismatches():
    for v in H["elide"]:
        assert [list(x) for x in pywfa_amd.elide_mismatches_from_cigar(_t(v["in"]))] == v["out"]
    assert pywfa_amd.elide_mismatches_from_cigar([]) == []


def test_clip_cigartuples():
    for v in H["clip"]:
        res = AlignmentResult(v["pl"], v["tl"], 0, v["pl"], v["ts0"], v["tl"], _t(v["ct"]), -7, "P" * v["pl"], "T" * v["tl"], 0)
        out = pywfa_amd.clip_cigartuples(res, v["left"], v["right"])
        assert out is res
        got = {"cigartuples": [list(x) for x in out.cigartuples], "text_start": out.text_start, "text_end": out.text_end,
               "pattern_start": out.pattern_start, "pattern_end": out.pattern_end}
        assert got == v["out"]


def test_alignment_result_views():
    ct = [(0, 3), (8, 1), (0, 4), (2, 1), (0, 7), (1, 1), (0, 9), (8, 1), (0, 6)]
    p, t = "TCTTTACTCGCGCGTTGGAGAAATACAATAGT", "TCTATACTGCGCGTTTGGAGAAATAAAATAGT"
    res = AlignmentResult(32, 32, 0, 32, 0, 32, ct, -24, p, t, 0)
    assert res.cigarstring == "3M1X4M1D7M1I9M1X6M"
    assert res.aligned_pattern == p and res.aligned_text == t
    pretty = res.pretty.splitlines()
    assert pretty[0].startswith("3M1X4M1D7M1I9M1X6M") and "PATTERN" in pretty[2]
    assert "Score: -24" in str(res)
    assert repr(res).startswith("    score: -24\n")
    empty = AlignmentResult(4, 0, 0, 0, 0, 0, [], -14, "", "", 0)
    assert empty.aligned_pattern is None and str(empty) == "Score: -14"
