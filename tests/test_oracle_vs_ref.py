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


@pytest.mark.parametrize("kw0", [dict(span="end-to-end"), dict(), dict(distance="affine2p"), dict(distance="levenshtein", span="end-to-end"),
                                 dict(distance="linear", mismatch=3, gap_extension=5), dict(match=-1, span="end-to-end"), dict(distance="indel")])
def test_biwfa_step_limit_equals_reference(kw0):
    """max_steps in BiWFA: the limit counts forward + reverse scores in the breakpoint search
    (R/wavefront_bialign.c:475,513 -> status -100, score unset) and the base cases carry it too (R/wavefront_bialigner.c:168-174:
    a base case that hits it is not "completed" -> -300); both scopes; status, score and the op string appended so far."""
    import validate_oracle as vo
    corpora = [datagen.generate(300, 150, 0.05, 11), datagen.generate(150, 150, 0.2, 12), datagen.generate(200, 60, 0.1, 13),
               datagen.generate(20, 1500, 0.08, 14), vo.corpus_special(seed=5)]
    for ms in (5, 60, 300, 3000):
        for scope in ("full", "score"):
            cfg = loader.make_config(**dict(kw0, scope=scope, memory_mode="biwfa", max_steps=ms))
            for batch in corpora:
                r = loader.run(loader.reference(), cfg, batch)
                o = loader.run(loader.oracle(), cfg, batch)
                common.assert_same(r, o["score"], o["status"], o["cigars"], batch, f"biwfa max_steps={ms} {scope} {kw0}")


BIWFA_HEUR = [dict(heuristic="adaptive"), dict(heuristic="adaptive", min_wavefront_length=5, max_distance_threshold=15, steps_between_cutoffs=3),
              dict(heuristic="X-drop", xdrop=400), dict(heuristic="X-drop", xdrop=100, match=-1), dict(heuristic="X-drop", xdrop=20),
              dict(heuristic="adaptive", distance="affine2p"), dict(heuristic="adaptive", distance="levenshtein"),
              dict(heuristic="adaptive", distance="linear", mismatch=3, gap_extension=5), dict(heuristic="adaptive", max_steps=400)]


@pytest.mark.parametrize("cfg_idx", range(len(BIWFA_HEUR)))
@pytest.mark.parametrize("scope", ["full", "score"])
def test_biwfa_with_a_heuristic_equals_reference(cfg_idx, scope):
    """Round 4: memory_mode="biwfa" with a heuristic.  The forward and the reverse aligner of every breakpoint search inherit it
    (R/wavefront_bialigner.c:53,161-166) and cut their wavefronts off after every extension (R/wavefront_extend.c:117-123,206-212),
    each with its own state, re-set at every search (R/wavefront_heuristic.c:114-121); the base cases run without
    (R/wavefront_bialigner.c:66-68).  The oracle's restatement against the real library in its ultralow mode: status, score, op string."""
    import validate_oracle as vo
    kw = dict(BIWFA_HEUR[cfg_idx], scope=scope, memory_mode="biwfa", span="end-to-end")
    corpora = [datagen.generate(n, L, e, 3900 + 7 * cfg_idx + i)
               for i, (n, L, e) in enumerate([(300, 150, 0.02), (150, 150, 0.2), (150, 60, 0.1), (30, 1500, 0.08), (10, 4000, 0.15), (4, 10000, 0.08)])]
    corpora.append(vo.corpus_special(seed=15 + cfg_idx))
    for batch in corpora:
        cfg = loader.make_config(**kw)
        r = loader.run(loader.reference(), cfg, batch)
        o = loader.run(loader.oracle(), cfg, batch)
        common.assert_same(r, o["score"], o["status"], o["cigars"], batch, f"biwfa + heuristic {kw}")


def test_biwfa_outside_the_built_subset_is_refused():
    batch = datagen.generate(4, 50, 0.05, 1)
    for kw in (dict(scope="score", span="ends-free", text_end_free=5), dict(scope="full", span="ends-free", pattern_begin_free=3)):
        with pytest.raises(Exception):
            loader.run(loader.oracle(), loader.make_config(**dict(kw, memory_mode="biwfa")), batch, want_cigar=False)


# ---- match < 0 with free begins: the ends-free re-seeding (R/wavefront_compute.c:124-254), SURVEY §8 f3 --------------------
# What can be pinned.  The reference leaves parts of a re-seeded wavefront unwritten: a null step whose begin-free cell exists
# on one side only gets lo = hi = +-j with wf_elements_init_min = init_max = 0 (R/wavefront.c:107-108), so later reads of the
# diagonals between 0 and j take whatever the slab held before — its scores then depend on the alignments the process ran
# earlier (tools/ref_endsfree_repro.py shows one pair scoring 72 alone and 73 after other pairs).  With pattern_begin_free ==
# text_begin_free both cells exist at every re-seeded null step, everything in between is written, and the library is
# deterministic: those rows are pinned here, in score scope (with a backtrace it exits or hangs, below).
EF_SEED = [dict(distance=d, match=m, mismatch=x, gap_opening=o, gap_extension=e, pattern_begin_free=f, text_begin_free=f,
                pattern_end_free=pe, text_end_free=te, heuristic=h)
           for d in ("affine", "affine2p", "linear")
           for (m, x, o, e) in ((-1, 3, 6, 2), (-2, 4, 6, 2), (-1, 4, 6, 2), (-3, 2, 1, 1))
           for (f, pe, te) in ((9, 7, 2), (12, 0, 0), (30, 5, 5))
           for h in (None, "adaptive")]


@pytest.mark.parametrize("cfg_idx", range(len(EF_SEED)))
def test_endsfree_reseeding_equals_reference_where_it_is_defined(cfg_idx):
    import validate_oracle as vo
    kw = dict(EF_SEED[cfg_idx], span="ends-free", scope="score")
    for batch in (datagen.generate(300, 150, 0.06, 4100 + cfg_idx), datagen.generate(12, 1000, 0.08, 4200 + cfg_idx), vo.corpus_special(seed=40 + cfg_idx)):
        kw2 = common.clamp_free(kw, batch)
        if kw2["pattern_begin_free"] != kw2["text_begin_free"]:
            kw2["pattern_begin_free"] = kw2["text_begin_free"] = min(kw2["pattern_begin_free"], kw2["text_begin_free"])
        cfg = loader.make_config(**kw2)
        loader.oracle_undefined_reads(True)
        o = loader.run(loader.oracle(), cfg, batch, want_cigar=False)
        if loader.oracle_undefined_reads():
            continue   # (free begins shorter than the first mismatch score: the reference's unset offsets[0], see the oracle)
        r = loader.run(loader.reference(), cfg, batch, want_cigar=False)
        assert np.array_equal(r["score"], o["score"]) and np.array_equal(r["status"], o["status"]), kw2


def test_reference_fails_with_a_backtrace_under_match_lt_0_and_free_begins():
    """Why scope=full is refused for match < 0 with free begins: on an ordinary 150 bp batch the real library exit(-1)s with
    "I?/D?-Beginning backtrace error" in memory mode high and does not return in memory mode medium (run in a child)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, {root!r})\n"
            "from oracle import loader\nfrom pywfa_amd import datagen\n"
            "cfg = loader.make_config(distance='affine', match=-1, mismatch=3, span='ends-free', pattern_begin_free=8, pattern_end_free=7,"
            " text_begin_free=3, text_end_free=2, scope='full', memory_mode={mem!r})\n"
            "loader.run(loader.reference(), cfg, datagen.generate(1500, 150, 0.06, 77))\nprint('survived')\n")
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    p = subprocess.run([sys.executable, "-c", code.format(root=root, mem="high")], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "backtrace error" in p.stderr and "survived" not in p.stdout
    try:
        p = subprocess.run([sys.executable, "-c", code.format(root=root, mem="medium")], capture_output=True, text=True, timeout=20)
        assert "survived" not in p.stdout   # (should it ever return, it has not produced a usable result)
    except subprocess.TimeoutExpired:
        pass
    with pytest.raises(Exception):   # the oracle refuses the configuration instead
        loader.run(loader.oracle(), loader.make_config(match=-1, pattern_begin_free=3, scope="full"), datagen.generate(2, 50, 0.05, 1))
