"""Where oracle/_ref (the real WFA2-lib, compiled from /root/reference) is present: the restatement
must agree with it bit-for-bit on fresh seeded corpora.  (tools/validate_oracle.py is the long form.)"""
import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import datagen

pytestmark = pytest.mark.skipif(not loader.have_reference(), reason="oracle/_ref not built (needs /root/reference)")

CONFIGS = [
    dict(span="end-to-end", scope="score"),
    dict(span="end-to-end", scope="full"),
    dict(scope="full"),
    dict(distance="affine2p", scope="full"),
    dict(distance="affine2p", span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2),
    dict(heuristic="adaptive"),
    dict(heuristic="X-drop", xdrop=100),
    dict(heuristic="X-drop", xdrop=20, scope="score"),
    dict(max_steps=10),
    dict(match=-1, span="end-to-end"),
    dict(distance="indel"), dict(distance="levenshtein", heuristic="adaptive"), dict(distance="linear", mismatch=3, gap_extension=5),
    dict(distance="linear", match=-1, span="end-to-end", scope="score"),
]


@pytest.mark.parametrize("cfg_idx", range(len(CONFIGS)))
@pytest.mark.parametrize("shape", [(3000, 150, 0.02), (600, 150, 0.15), (60, 1000, 0.08)])
def test_oracle_equals_reference(cfg_idx, shape):
    n, L, e = shape
    batch = datagen.generate(n, L, e, 900 + cfg_idx)
    kw = common.clamp_free(CONFIGS[cfg_idx], batch)
    cfg = loader.make_config(**kw)
    r = loader.run(loader.reference(), cfg, batch)
    o = loader.run(loader.oracle(), cfg, batch)
    common.assert_same(r, o["score"], o["status"], o["cigars"], batch, f"oracle vs reference {kw}")


BIWFA = [dict(span="end-to-end"), dict(), dict(distance="affine2p", span="end-to-end"), dict(distance="indel"),
         dict(distance="levenshtein", span="end-to-end"), dict(distance="linear", mismatch=3, gap_extension=5),
         dict(match=-1, span="end-to-end"), dict(wildcard="N")]


@pytest.mark.parametrize("cfg_idx", range(len(BIWFA)))
def test_biwfa_score_scope_equals_reference(cfg_idx):
    """memory_mode="biwfa" is built for scope=score without heuristic / free ends / step limit: the
    oracle must return what the real library returns in its ultralow mode, which is also what its high mode returns."""
    kw = dict(BIWFA[cfg_idx], scope="score", memory_mode="biwfa")
    for i, (n, L, e) in enumerate([(2000, 150, 0.02), (400, 150, 0.2), (40, 1200, 0.08), (1500, 40, 0.1)]):
        batch = datagen.generate(n, L, e, 1900 + 7 * cfg_idx + i)
        cfg = loader.make_config(**kw)
        r = loader.run(loader.reference(), cfg, batch, want_cigar=False)
        o = loader.run(loader.oracle(), cfg, batch, want_cigar=False)
        h = loader.run(loader.reference(), loader.make_config(**dict(kw, memory_mode="high")), batch, want_cigar=False)
        assert np.array_equal(r["score"], o["score"]) and np.array_equal(r["status"], o["status"]), kw
        assert np.array_equal(r["score"], h["score"]) and np.array_equal(r["status"], h["status"]), kw


BIWFA_FULL = [dict(span="end-to-end"), dict(), dict(distance="affine2p"), dict(distance="affine2p", span="end-to-end", mismatch=3, gap_opening=4, gap_extension=2, gap_opening2=12, gap_extension2=1),
              dict(distance="indel"), dict(distance="levenshtein", span="end-to-end"), dict(distance="linear", mismatch=3, gap_extension=5),
              dict(match=-1, span="end-to-end"), dict(wildcard="N"), dict(mismatch=2, gap_opening=3, gap_extension=1)]


@pytest.mark.parametrize("cfg_idx", range(len(BIWFA_FULL)))
def test_biwfa_full_cigar_equals_reference(cfg_idx):
    """memory_mode="biwfa", scope=full: the breakpoint recursion of R/wavefront_bialign.c restated in the oracle against the
    real library in its ultralow mode — status, score (incl. the unset INT32_MIN score when the top level is answered by
    the ordinary algorithm) and op string, from reads that never split (<= 100 bases) to reads that
    split several levels deep (score >> 250)."""
    import validate_oracle as vo
    kw = dict(BIWFA_FULL[cfg_idx], scope="full", memory_mode="biwfa")
    corpora = [datagen.generate(n, L, e, 2900 + 7 * cfg_idx + i)
               for i, (n, L, e) in enumerate([(600, 150, 0.02), (200, 150, 0.2), (300, 60, 0.1), (40, 1500, 0.08), (12, 4000, 0.15), (6, 10000, 0.08)])]
    corpora.append(vo.corpus_special(seed=5 + cfg_idx))
    for batch in corpora:
        cfg = loader.make_config(**kw)
        r = loader.run(loader.reference(), cfg, batch)
        o = loader.run(loader.oracle(), cfg, batch)
        common.assert_same(r, o["score"], o["status"], o["cigars"], batch, f"biwfa full {kw}")
        # same optimal score as the ordinary algorithm wherever BiWFA reports one
        h = loader.run(loader.reference(), loader.make_config(**dict(kw, memory_mode="high")), batch)
        have = o["score"] != -2147483648
        assert np.array_equal(o["score"][have], h["score"][have]), kw


def test_biwfa_outside_the_built_subset_is_refused():
    batch = datagen.generate(4, 50, 0.05, 1)
    for kw in (dict(scope="full", heuristic="adaptive"), dict(scope="score", heuristic="adaptive"), dict(scope="score", max_steps=50),
               dict(scope="score", span="ends-free", text_end_free=5), dict(scope="full", span="ends-free", pattern_begin_free=3)):
        with pytest.raises(Exception):
            loader.run(loader.oracle(), loader.make_config(**dict(kw, memory_mode="biwfa")), batch, want_cigar=False)
