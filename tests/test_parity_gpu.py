"""Parity tests proper: the HIP path (through the C ABI of libwfa_hip.so) against the oracle on the
same seeded inputs, and against the committed golden vectors.  Bit-exact: status, score, op string."""
import os

import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import datagen

pytestmark = pytest.mark.gpu

C_LEVEL = common.load_golden("c_level.json")


@pytest.mark.parametrize("run_idx", range(len(C_LEVEL["runs"])))
def test_hip_matches_golden_vectors(gpu, run_idx):
    run = C_LEVEL["runs"][run_idx]
    pairs = C_LEVEL["corpora"][run["corpus"]]
    batch = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
    oc, nc = common.configs_pair(**run["config"])
    full = oc.scope == 1
    score, status, cigars = common.gpu_run(nc, batch, full, resident=(run_idx % 2 == 0))
    assert score.tolist() == run["score"]
    assert status.tolist() == run["status"]
    if run["cigar"] is not None:
        assert [common.rle(c) for c in cigars] == run["cigar"]


GRID = []
for distance in ("affine", "affine2p"):
    for span, free in (("end-to-end", (0, 0, 0, 0)), ("ends-free", (0, 0, 0, 0)), ("ends-free", (8, 7, 3, 2)), ("ends-free", (20, 0, 0, 9))):
        for heur in (None, "adaptive", ("X-drop", 20), ("X-drop", 100)):
            for scope in ("score", "full"):
                kw = dict(distance=distance, span=span, scope=scope, pattern_begin_free=free[0], pattern_end_free=free[1],
                          text_begin_free=free[2], text_end_free=free[3])
                if isinstance(heur, tuple):
                    kw.update(heuristic=heur[0], xdrop=heur[1])
                else:
                    kw.update(heuristic=heur)
                GRID.append(kw)
GRID += [
    dict(distance="affine", mismatch=2, gap_opening=3, gap_extension=1),
    dict(distance="affine", mismatch=5, gap_opening=0, gap_extension=3),
    dict(distance="affine2p", mismatch=3, gap_opening=4, gap_extension=2, gap_opening2=12, gap_extension2=1),
    dict(distance="affine", match=-1, span="end-to-end"),
    dict(distance="affine2p", match=-1, span="ends-free", pattern_end_free=5, text_end_free=5),
    dict(distance="affine", match=-1, heuristic="X-drop", xdrop=100, span="end-to-end", scope="score"),
    dict(distance="affine", heuristic="adaptive", min_wavefront_length=5, max_distance_threshold=10, steps_between_cutoffs=3),
    dict(distance="affine", max_steps=10), dict(distance="affine2p", max_steps=25, scope="score"),
    dict(distance="affine", memory_mode="medium"), dict(distance="affine2p", memory_mode="low", span="end-to-end"),
    # BiWFA, the built subset: score scope, no heuristic / free ends / step limit
    dict(distance="affine", memory_mode="biwfa", scope="score", span="end-to-end"), dict(distance="affine2p", memory_mode="biwfa", scope="score"),
    dict(distance="levenshtein", memory_mode="biwfa", scope="score", span="end-to-end"),
    dict(distance="affine", wildcard="N"),
    # single-component metrics
    dict(distance="indel"), dict(distance="indel", span="end-to-end", scope="score"),
    dict(distance="levenshtein"), dict(distance="levenshtein", span="end-to-end", scope="score", heuristic="adaptive"),
    dict(distance="levenshtein", span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2),
    dict(distance="linear"), dict(distance="linear", mismatch=3, gap_extension=5, span="end-to-end"),
    dict(distance="linear", heuristic="X-drop", xdrop=100), dict(distance="linear", match=-1, span="end-to-end"),
    dict(distance="linear", max_steps=12, scope="score"), dict(distance="indel", heuristic="adaptive", scope="score"),
    # match < 0 with free begins: the ends-free re-seeding (R/wavefront_compute.c:124-254), score scope.  Equal free begins: the
    # rows the oracle is pinned on against the real library (tests/test_oracle_vs_ref.py); unequal: the reference reads cells it
    # never wrote there (tools/ref_endsfree_repro.py), the oracle's NULL reading is what is compared
    dict(distance="affine", match=-1, mismatch=3, span="ends-free", pattern_begin_free=9, pattern_end_free=7, text_begin_free=9, text_end_free=2, scope="score"),
    dict(distance="affine2p", match=-2, span="ends-free", pattern_begin_free=12, text_begin_free=12, scope="score", heuristic="adaptive"),
    dict(distance="linear", match=-1, mismatch=3, gap_extension=2, span="ends-free", pattern_begin_free=9, text_begin_free=9, pattern_end_free=3, scope="score"),
    dict(distance="affine", match=-1, span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2, scope="score"),
    dict(distance="affine2p", match=-1, mismatch=3, span="ends-free", pattern_begin_free=20, text_end_free=9, scope="score", heuristic="X-drop", xdrop=100),
]

SHAPES = {"150bp_2pct": (1500, 150, 0.02), "150bp_15pct": (400, 150, 0.15), "1kb_8pct": (60, 1000, 0.08)}


@pytest.fixture(scope="module")
def corpora():
    import validate_oracle as vo
    out = {name: datagen.generate(n, L, e, 7000 + L + int(e * 1000)) for name, (n, L, e) in SHAPES.items()}
    out["special"] = vo.corpus_special(seed=99)   # empty / length-1 / N / repeats / long gaps / windows
    # (every configuration meets the edge cases whatever the sampling below skips — 64 pairs of the special corpus: the
    # empty / length-1 / identical / unrelated block, then homopolymers, two-letter reads, reads with N, long gaps, windows)
    sp = out["special"]
    n_sp = len(sp["p_len"])
    pick = list(range(27)) + [27 + 30 * i for i in range(10)] + [327 + 33 * i for i in range(6)] + [527 + 12 * i for i in range(8)] + \
           [627 + 25 * i for i in range(8)] + [827 + 40 * i for i in range(5)]
    pick = [i for i in pick if i < n_sp]
    pats = [bytes(sp["seqs"][sp["p_off"][i]:sp["p_off"][i] + sp["p_len"][i]]).decode() for i in pick]
    txts = [bytes(sp["seqs"][sp["t_off"][i]:sp["t_off"][i] + sp["t_len"][i]]).decode() for i in pick]
    out["special_mini"] = datagen.from_strings(pats, txts)
    return out


# which two of the four large corpora a configuration runs on under -m gpu: the six possible pairs in turn, advanced so that neither
# `scope` (alternates with cfg_idx), the heuristic (every 2), the span (every 8) nor the distance (every 32) is tied to a corpus
# (the old stride (cfg_idx + corpus_idx) % 2 ran scope=score on two corpora only and scope=full on the other two)
CORPUS_PAIRS = [(0, 1), (2, 3), (0, 2), (1, 3), (0, 3), (1, 2)]


def sampled(cfg_idx, corpus_idx):
    return corpus_idx in CORPUS_PAIRS[(cfg_idx + cfg_idx // 2 + cfg_idx // 8 + cfg_idx // 32) % 6]


@pytest.mark.parametrize("cfg_idx", range(len(GRID)))
@pytest.mark.parametrize("corpus", list(SHAPES) + ["special", "special_mini"])
def test_hip_matches_oracle(gpu, corpora, corpus, cfg_idx, monkeypatch):
    # (the 4-corpus x 90-configuration grid is sampled under -m gpu — every configuration on two of the four large
    # corpora (CORPUS_PAIRS) and always on the 64 edge-case pairs of `special_mini`; WFA_TEST_FULL=1 runs all of it: the log of such a
    # run on the round's final kernels is profiles/r06_gputests_full.txt)
    if os.environ.get("WFA_TEST_FULL") != "1" and corpus != "special_mini" and not sampled(cfg_idx, (list(SHAPES) + ["special"]).index(corpus)):
        pytest.skip("grid sampled (WFA_TEST_FULL=1 runs every cell)")
    # (non-resident calls of <= 4 096 short pairs take the single-launch path; every fourth configuration keeps the batch machinery —
    # pageable upload, device pack, the kernel cascade — covered at these sizes)
    if cfg_idx % 4 == 0:
        monkeypatch.setenv("WFA_HIP_NO_TINY", "1")
    batch = corpora[corpus]
    kw = common.clamp_free(GRID[cfg_idx], batch)
    oc, nc = common.configs_pair(**kw)
    o = loader.run(loader.oracle(), oc, batch)
    full = oc.scope == 1
    score, status, cigars = common.gpu_run(nc, batch, full, resident=(cfg_idx % 2 == 1))
    common.assert_same(o, score, status, cigars, batch, f"{corpus} {kw}")


@pytest.mark.parametrize("kw", [dict(span="end-to-end", scope="full", heuristic="adaptive"),
                                dict(span="end-to-end", scope="score"),
                                dict(distance="affine2p", span="ends-free", pattern_end_free=100, text_end_free=100, heuristic="adaptive")])
def test_hip_matches_oracle_10kb(gpu, kw):
    batch = datagen.generate(24, 10000, 0.08, 1003)
    oc, nc = common.configs_pair(**kw)
    o = loader.run(loader.oracle(), oc, batch)
    score, status, cigars = common.gpu_run(nc, batch, oc.scope == 1, resident=True)
    common.assert_same(o, score, status, cigars, batch, f"10kb {kw}")


def test_exact_full_cigar_needs_arena_growth(gpu):
    """Exact (no heuristic) full-CIGAR alignment of 4 kb reads stores ~10 MB of wavefront history per
    pair: the first arena overflows and the pair is re-run with a larger one; results are unchanged."""
    batch = datagen.generate(6, 4000, 0.08, 31)
    oc, nc = common.configs_pair(span="end-to-end", scope="full")
    o = loader.run(loader.oracle(), oc, batch)
    score, status, cigars = common.gpu_run(nc, batch, True, resident=True)
    common.assert_same(o, score, status, cigars, batch, "4kb exact full")


def test_ragged_and_empty_batches(gpu):
    from pywfa_amd import _native
    oc, nc = common.configs_pair(scope="full")
    al = _native.Aligner(nc)
    empty = datagen.from_strings([], [])
    score, status, cig = al.align_batch(empty, True)
    assert score.size == 0 and status.size == 0
    ragged = datagen.from_strings(["", "A", "ACGT" * 100, "ACGTTTGA"], ["ACGT", "", "ACGT" * 90 + "TT", "ACGTTTGA"])
    o = loader.run(loader.oracle(), oc, ragged)
    score, status, cig = al.align_batch(ragged, True)
    ops, cbeg, clen = cig
    cigars = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in range(4)]
    common.assert_same(o, score, status, cigars, ragged, "ragged")
    assert common.rle(cigars[0]) == "4I" and score[0] == -14   # empty pattern
    al.close()


LANE_GENERAL = [dict(span="end-to-end", heuristic="adaptive"), dict(span="ends-free", heuristic="adaptive"),
                dict(span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2),
                dict(span="ends-free", pattern_begin_free=4, pattern_end_free=0, text_begin_free=2, text_end_free=9, heuristic="adaptive"),
                dict(span="ends-free", pattern_begin_free=20, text_end_free=9), dict(span="end-to-end", max_steps=10), dict(max_steps=24, heuristic="adaptive"),
                dict(span="end-to-end", heuristic="adaptive", min_wavefront_length=5, max_distance_threshold=10, steps_between_cutoffs=3),
                dict(span="end-to-end", heuristic="adaptive", mismatch=2, gap_opening=3, gap_extension=1),
                dict(span="ends-free", text_begin_free=5, pattern_end_free=5, mismatch=3, gap_opening=3, gap_extension=1, heuristic="adaptive", min_wavefront_length=3)]


@pytest.mark.parametrize("slots", [16, 32])
@pytest.mark.parametrize("cfg_idx", range(len(LANE_GENERAL)))
def test_lane_kernel_general_form(gpu, corpora, cfg_idx, slots, monkeypatch):
    """The general score-only form of the lane kernel (wfa_lane_kernel<.., HEUR>, round 3: wf-adaptive, free ends and the step limit
    under the no-clipping rule) forced on (WFA_HIP_LANE_HEUR=1: small batches skip its pilot; 2: its 32-diagonal form of round 6), over
    every corpus: what it keeps and what it hands on to the banded stages must equal the oracle."""
    if slots == 32 and cfg_idx % 2 == 1 and cfg_idx != 4 and os.environ.get("WFA_TEST_FULL") != "1":
        pytest.skip("sampled on the suite's time budget (WFA_TEST_FULL=1 runs every cell)")   # (4: free begins of 20 diagonals fit the 32 slots only)
    monkeypatch.setenv("WFA_HIP_LANE_HEUR", "1" if slots == 16 else "2")
    for name in ("150bp_2pct", "150bp_15pct", "special"):
        batch = corpora[name]
        kw = common.clamp_free(dict(LANE_GENERAL[cfg_idx], scope="score"), batch)
        oc, nc = common.configs_pair(**kw)
        o = loader.run(loader.oracle(), oc, batch, want_cigar=False)
        score, status, _ = common.gpu_run(nc, batch, False, resident=(cfg_idx % 2 == 1))
        common.assert_same(o, score, status, None, batch, f"lane general form {name} {kw}")


@pytest.mark.parametrize("lane_first", [0, 1, 2])
@pytest.mark.parametrize("cfg_idx", [i for i, c in enumerate(LANE_GENERAL) if "max_steps" not in c])
def test_segmented_kernel_general_form(gpu, corpora, cfg_idx, lane_first, monkeypatch):
    """The same form of the 32-lane segments (wfa_seg_kernel<.., 32, .., HEUR>: two pairs per wave, a band of 32 diagonals) forced on
    (WFA_HIP_SEG_HEUR=1), alone and behind the lane form: what it keeps and what it hands on must equal the oracle."""
    if lane_first == 2 and (cfg_idx % 2 == 0 or cfg_idx == 9) and os.environ.get("WFA_TEST_FULL") != "1":   # (9: a run-time shape: seconds of compiling)
        pytest.skip("sampled on the suite's time budget (WFA_TEST_FULL=1 runs every cell)")
    monkeypatch.setenv("WFA_HIP_SEG_HEUR", "1")
    monkeypatch.setenv("WFA_HIP_LANE_HEUR", str(lane_first))
    for name in ("150bp_2pct", "150bp_15pct", "special"):
        batch = corpora[name]
        kw = common.clamp_free(dict(LANE_GENERAL[cfg_idx], scope="score"), batch)
        oc, nc = common.configs_pair(**kw)
        o = loader.run(loader.oracle(), oc, batch, want_cigar=False)
        score, status, _ = common.gpu_run(nc, batch, False, resident=(cfg_idx % 2 == 0))
        common.assert_same(o, score, status, None, batch, f"segment general form {name} {kw}")


SEG_XDROP = [dict(span="end-to-end", heuristic="X-drop", xdrop=100), dict(span="end-to-end", heuristic="X-drop", xdrop=20),
             dict(span="end-to-end", heuristic="X-drop", xdrop=8, steps_between_cutoffs=3),
             dict(span="ends-free", heuristic="X-drop", xdrop=30, pattern_begin_free=4, pattern_end_free=6, text_begin_free=2, text_end_free=9),
             dict(span="end-to-end", heuristic="X-drop", xdrop=15, mismatch=2, gap_opening=3, gap_extension=1)]


@pytest.mark.parametrize("cfg_idx", range(len(SEG_XDROP)))
def test_segmented_kernel_general_form_xdrop(gpu, corpora, cfg_idx, monkeypatch):
    """Round 4: X-drop (R/wavefront_heuristic.c:297-383) in the general form of the 32-lane segments, forced on: scores and statuses
    (a dropped alignment ends unreachable in the banded stage it is handed on to) must equal the oracle's."""
    monkeypatch.setenv("WFA_HIP_SEG_HEUR", "1")
    for name in ("150bp_2pct", "150bp_15pct", "special"):
        batch = corpora[name]
        kw = common.clamp_free(dict(SEG_XDROP[cfg_idx], scope="score"), batch)
        oc, nc = common.configs_pair(**kw)
        o = loader.run(loader.oracle(), oc, batch, want_cigar=False)
        score, status, _ = common.gpu_run(nc, batch, False, resident=(cfg_idx % 2 == 0))
        common.assert_same(o, score, status, None, batch, f"segment general form, X-drop {name} {kw}")


@pytest.mark.parametrize("xdrop", [100, 20])
@pytest.mark.parametrize("error", [0.005, 0.02])
def test_segmented_kernel_general_form_xdrop_pilot(gpu, error, xdrop):
    """A batch large enough for the pilot (>= 64 k pairs): X-drop(100) takes the segmented form at 0.5 % divergence (few pairs outgrow
    its 32 diagonals) and the banded kernel alone at 2 % (a fifth would be handed on); X-drop(20) drops every cell of many alignments,
    which then end "unreachable" inside the segmented form (round 6); either way the results are the real library's."""
    batch = datagen.generate(70000, 150, error, 4456)
    oc, nc = common.configs_pair(span="end-to-end", scope="score", heuristic="X-drop", xdrop=xdrop)
    o = loader.run(loader.reference() if loader.have_reference() else loader.oracle(), oc, batch, want_cigar=False)
    score, status, _ = common.gpu_run(nc, batch, False, resident=True)
    common.assert_same(o, score, status, None, batch, f"segment general form, X-drop, pilot {error}")


@pytest.mark.parametrize("error", [0.005, 0.02, 0.03])
def test_lane_kernel_general_form_pilot(gpu, error):
    """Batches of >= 64 k pairs let a pilot decide which general form goes first (few pairs outgrow the lane kernel's 16 slots at 0.5 %
    divergence; at 2 % two fifths do and 2 % its 32 slots: the 32-slot form; at 3 % a tenth outgrow those too: the 32-lane segments);
    whichever, the results are the oracle's."""
    batch = datagen.generate(70000, 150, error, 4455)
    oc, nc = common.configs_pair(span="end-to-end", scope="score", heuristic="adaptive")
    o = loader.run(loader.reference() if loader.have_reference() else loader.oracle(), oc, batch, want_cigar=False)
    score, status, _ = common.gpu_run(nc, batch, False, resident=True)
    common.assert_same(o, score, status, None, batch, f"lane general form pilot {error}")


def test_match_lt_0_with_free_begins_and_a_backtrace_is_refused(gpu):
    """With a backtrace the reference itself exits or hangs on this configuration (tests/test_oracle_vs_ref.py): NotImplementedError."""
    from pywfa_amd import _native
    c = _native.default_config()
    c.match, c.pattern_begin_free = -1, 4
    assert _native.validate(c)[0] == _native.ENOTSUP
    c.scope = 0
    assert _native.validate(c)[0] == _native.OK
    import pywfa_amd
    with pytest.raises(NotImplementedError):
        pywfa_amd.WavefrontAligner("ACGT", match=-1, pattern_begin_free=2)
    a = pywfa_amd.WavefrontAligner("ACGTTTGACA", match=-1, mismatch=3, pattern_begin_free=3, text_begin_free=3, scope="score")
    cfg = loader.make_config(match=-1, mismatch=3, pattern_begin_free=3, text_begin_free=3, scope="score")
    o = loader.run(loader.oracle(), cfg, datagen.from_strings(["ACGTTTGACA"], ["TTACGTTAGACA"]), want_cigar=False)
    assert a.wavefront_align("TTACGTTAGACA") == int(o["score"][0])
    a.close()


def test_ends_free_larger_than_sequence_is_an_error(gpu):
    """The reference exit(1)s (wavefront_align.c:95-101); here: ValueError."""
    from pywfa_amd import _native
    oc, nc = common.configs_pair(pattern_begin_free=50)
    al = _native.Aligner(nc)
    with pytest.raises(ValueError):
        al.align_batch(datagen.from_strings(["ACGT"], ["ACGT"]), True)
    al.close()


def test_mixed_lengths_and_alphabets_one_batch(gpu):
    """One batch mixing 20 bp .. 3 kb pairs, pure-ACGT and N-containing pairs: the cascade (two-per-wave,
    one-per-wave, banded, general, 8-bit) must hand every pair to a stage that finishes it."""
    rng = np.random.default_rng(3)
    pats, txts = [], []
    for i in range(600):
        L = int(rng.choice([20, 60, 150, 150, 150, 300, 700, 3000]))
        b = datagen.generate(1, L, float(rng.choice([0.0, 0.02, 0.1])), 10_000 + i)
        p, t = datagen.pair_strings(b, 0)
        if i % 7 == 0:
            p = p[: L // 2] + "N" + p[L // 2 + 1:]
        pats.append(p); txts.append(t)
    batch = datagen.from_strings(pats, txts)
    for kw in (dict(span="end-to-end", scope="score"), dict(scope="full"), dict(scope="full", heuristic="adaptive")):
        oc, nc = common.configs_pair(**kw)
        o = loader.run(loader.oracle(), oc, batch)
        for resident in (False, True):
            score, status, cigars = common.gpu_run(nc, batch, oc.scope == 1, resident)
            common.assert_same(o, score, status, cigars, batch, f"mixed {kw}")


def test_resident_batch_reruns_are_identical(gpu):
    from pywfa_amd import _native
    batch = datagen.generate(5000, 1500, 0.08, 77)
    oc, nc = common.configs_pair(span="end-to-end", scope="full", heuristic="adaptive")
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    outs = []
    for _ in range(3):
        rb.run(); rb.run(); rb.sync()          # two enqueued runs, then a sync
        s, st, (ops, cb, cl) = rb.results(True)
        outs.append((s.copy(), st.copy(), [ops[cb[i]:cb[i] + cl[i]].tobytes() for i in range(0, 5000, 97)]))
    rb.close(); al.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and o[2] == outs[0][2]
    ref = loader.run(loader.oracle(), oc, datagen.subset(batch, np.arange(0, 5000, 97)))
    assert [c for c in ref["cigars"]] == outs[0][2]


@pytest.mark.parametrize("stages", ["689", "245", "2", "3", "4", "5", "6", "7", "8", "9", "76", "98", "3245", "1", "189", "12", "19"])
def test_segmented_stages_accept_only_proven_scores(gpu, stages, monkeypatch):
    """The banded segments (wfa_seg.hpp) keep a score only when it is provably the unbanded optimum and hand the
    pair on otherwise: whatever the stage order, the scores must be the reference's.  The batch mixes pairs
    that finish in every band width (0 % .. 30 % divergence, 30 .. 500 bases, |tlen - plen| up to 40), is long
    enough for a wave's slice to outrun its two metadata windows (> 128 pairs per wave) and has repeats."""
    if os.environ.get("WFA_TEST_FULL") != "1" and stages in ("4", "5", "7", "3245", "189", "19", "12", "245"):
        pytest.skip("stage orders sampled on the suite's time budget (WFA_TEST_FULL=1 runs all)")
    monkeypatch.setenv("WFA_HIP_FAST_STAGES", stages)
    monkeypatch.setenv("WFA_HIP_FAST_WAVES_PER_CU", "1")   # 256 waves -> ~200 pairs per slice
    rng = np.random.default_rng(11)
    parts = []
    for i, (L, e) in enumerate([(150, 0.0), (150, 0.02), (150, 0.06), (150, 0.15), (60, 0.3), (30, 0.1),
                                (500, 0.01), (500, 0.05), (330, 0.02), (16, 0.0)]):
        parts.append(datagen.generate(5000, L, e, 4000 + i))
    pats, txts = [], []
    for b in parts:
        for j in rng.choice(5000, 40, replace=False):
            p, t = datagen.pair_strings(b, int(j))
            cut = int(rng.integers(0, 41))
            pats.append(p[: max(1, len(p) - cut)] if j % 3 == 0 else p)   # end-to-end with a length difference
            txts.append(t)
    small = datagen.from_strings(pats, txts)
    oc, nc = common.configs_pair(span="end-to-end", scope="score")
    for batch in parts[:4] + [small]:
        o = loader.run(loader.oracle(), oc, batch, want_cigar=False)
        score, status, _ = common.gpu_run(nc, batch, False, True)
        assert np.array_equal(status, o["status"]), stages
        assert np.array_equal(score, o["score"]), stages
    # order of the pairs in the batch must not matter
    perm = rng.permutation(len(pats))
    shuf = datagen.from_strings([pats[i] for i in perm], [txts[i] for i in perm])
    o = loader.run(loader.oracle(), oc, small, want_cigar=False)
    score, status, _ = common.gpu_run(nc, shuf, False, False)
    assert np.array_equal(score, o["score"][perm]) and np.array_equal(status, o["status"][perm])


@pytest.mark.parametrize("pen", [(4, 4, 2), (4, 6, 1), (3, 4, 1), (6, 5, 3), (5, 0, 3), (1, 1, 1), (8, 12, 4), (2, 3, 1), (7, 11, 3)])
def test_segmented_kernel_penalty_shapes(gpu, pen):
    """Score-only end-to-end batches under the penalty shapes the segmented kernel is instantiated for (and one it is
    not: 7/11/3 runs in the general kernel) against the oracle, short and long slices, low and high divergence."""
    x, o, e = pen
    oc, nc = common.configs_pair(span="end-to-end", scope="score", mismatch=x, gap_opening=o, gap_extension=e)
    for i, (n, L, err) in enumerate([(6000, 150, 0.02), (3000, 150, 0.08), (1500, 150, 0.2), (1500, 480, 0.03), (800, 40, 0.1)]):
        batch = datagen.generate(n, L, err, 7000 + i)
        o_ = loader.run(loader.oracle(), oc, batch, want_cigar=False)
        score, status, _ = common.gpu_run(nc, batch, False, i % 2 == 0)
        assert np.array_equal(status, o_["status"]), (pen, L, err)
        assert np.array_equal(score, o_["score"]), (pen, L, err)


@pytest.mark.parametrize("span", ["ends-free", "end-to-end"])
def test_segmented_full_cigar_in_several_launches(gpu, span, monkeypatch):
    """Full CIGARs of short reads come from the lane kernel's full-CIGAR form (round 3; WFA_HIP_LANE_FULL=0: from the 16-lane segments
    with a history slot per pair); with a small slot budget the batch takes several launches over one region (work_begin > 0), some
    pairs exceed the band's bound and are finished by the stages behind, and a re-run of the resident batch gives the same op
    strings.  Also: the batch cut into launches with regions of their own (walks and expands on the side stream), and the default."""
    monkeypatch.setenv("WFA_HIP_SEGFULL_PAIRS", "1500")
    parts = [datagen.generate(4000, 150, 0.02, 9100), datagen.generate(1200, 150, 0.12, 9101), datagen.generate(1000, 400, 0.02, 9102),
             datagen.generate(700, 33, 0.06, 9103)]
    oc, nc = common.configs_pair(span=span, scope="full")
    for batch in parts:
        o = loader.run(loader.oracle(), oc, batch)
        for resident in (False, True):
            score, status, cigars = common.gpu_run(nc, batch, True, resident)
            common.assert_same(o, score, status, cigars, batch, f"segmented full {span}")
    monkeypatch.setenv("WFA_HIP_LANE_FULL", "0")
    for batch in parts[:2]:
        o = loader.run(loader.oracle(), oc, batch)
        score, status, cigars = common.gpu_run(nc, batch, True, True)
        common.assert_same(o, score, status, cigars, batch, f"segmented full {span}, 16-lane segments first")
    monkeypatch.delenv("WFA_HIP_LANE_FULL", raising=False)
    # the same through the default budget, in three launches with regions of their own, and without the stage
    for env in ({}, {"WFA_HIP_LANE_FULL_SPLIT": "3"}, {"WFA_HIP_NO_SEGFULL": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        monkeypatch.delenv("WFA_HIP_SEGFULL_PAIRS", raising=False)
        o = loader.run(loader.oracle(), oc, parts[0])
        score, status, cigars = common.gpu_run(nc, parts[0], True, True)
        common.assert_same(o, score, status, cigars, parts[0], f"segmented full {span} {env}")


@pytest.mark.parametrize("kw", [dict(span="end-to-end", memory_mode="medium"), dict(span="end-to-end", memory_mode="low", heuristic="adaptive"),
                                dict(memory_mode="medium"), dict(memory_mode="low", span="ends-free", pattern_begin_free=40, pattern_end_free=30,
                                                                 text_begin_free=25, text_end_free=35, heuristic="adaptive")])
@pytest.mark.parametrize("band_pb", ["default", "0"])
def test_piggyback_history_gives_the_same_cigars(gpu, kw, band_pb, monkeypatch):
    """Long reads keep one byte of origin codes per (step, diagonal) instead of the offsets and the op string is unpacked by
    re-extending the matches — by default in every memory mode; WFA_HIP_BAND_PB=0 keeps the explicit offsets.
    The reference returns the same alignments in all its memory modes, and so must both forms."""
    if band_pb != "default":
        monkeypatch.setenv("WFA_HIP_BAND_PB", band_pb)
    # (sizes on the suite's time budget, VERDICT r04 item 6: the oracle's CPU time is what these tests take)
    for i, (n, L, e) in enumerate([(300, 1500, 0.06), (100, 4000, 0.08), (32, 10000, 0.08), (150, 2500, 0.01)]):
        batch = datagen.generate(n, L, e, 8800 + i)
        kw2 = common.clamp_free(dict(kw, scope="full"), batch)
        oc, nc = common.configs_pair(**kw2)
        o = loader.run(loader.oracle(), oc, batch)
        oh = loader.run(loader.oracle(), common.configs_pair(**dict(kw2, memory_mode="high"))[0], batch)
        assert o["cigars"] == oh["cigars"]
        score, status, cigars = common.gpu_run(nc, batch, True, i % 2 == 0)
        common.assert_same(o, score, status, cigars, batch, f"piggy-back {kw2} L={L}")


@pytest.mark.parametrize("kw", [dict(heuristic="adaptive", memory_mode="medium"), dict(span="end-to-end", heuristic="adaptive", memory_mode="low"),
                                dict(memory_mode="low", span="ends-free", pattern_begin_free=40, pattern_end_free=30, text_begin_free=25,
                                     text_end_free=35, heuristic="adaptive"),
                                dict(memory_mode="medium", span="end-to-end")])
@pytest.mark.parametrize("band_pb", ["default", "0"])
def test_piggyback_history_gap_affine_2p(gpu, kw, band_pb, monkeypatch):
    """(band_pb = "0": the same inputs with the explicit 16-byte history entries.)
    The piggy-back history of the banded kernel for gap-affine-2p: seven bits of origin codes per (step,
    diagonal) — which of mismatch / D1 / D2 / I1 / I2 made M, and open-or-extend for each of I1, D1, I2, D2 — walked back
    with the reference's candidate priority (R/wavefront_backtrace.c:49-59).  Inputs with long gaps make the second gap
    piece win; the exact form (last case) runs 2p wavefronts past the 256-diagonal window into the general kernel."""
    if band_pb != "default":
        monkeypatch.setenv("WFA_HIP_BAND_PB", band_pb)
    exact = "heuristic" not in kw
    shapes = [(300, 1500, 0.06), (24, 3000, 0.08)] if exact else [(200, 1500, 0.06), (60, 4000, 0.08), (24, 10000, 0.08), (100, 2500, 0.01)]
    for i, (n, L, e) in enumerate(shapes):
        batch = datagen.generate(n, L, e, 8900 + i)
        if i == 0:
            # long gaps: cut 15-60 bases out of every second text
            rng = np.random.default_rng(5)
            pats, txts = [], []
            for j in range(n):
                p_, t_ = datagen.pair_strings(batch, j)
                if j % 2 == 0:
                    a_ = int(rng.integers(100, L - 200)); g_ = int(rng.integers(15, 61))
                    t_ = t_[:a_] + t_[a_ + g_:]
                pats.append(p_); txts.append(t_)
            batch = datagen.from_strings(pats, txts)
        kw2 = common.clamp_free(dict(kw, distance="affine2p", scope="full"), batch)
        oc, nc = common.configs_pair(**kw2)
        o = loader.run(loader.oracle(), oc, batch)
        score, status, cigars = common.gpu_run(nc, batch, True, i % 2 == 0)
        common.assert_same(o, score, status, cigars, batch, f"piggy-back 2p {kw2} L={L}")


@pytest.mark.parametrize("pen", [(4, 4, 2), (4, 6, 1), (3, 4, 1), (2, 3, 1)])
def test_banded_kernel_penalty_shapes(gpu, pen):
    """Long reads and full CIGARs under the penalty shapes the banded kernel is instantiated for: exact and wf-adaptive,
    explicit and piggy-back history, ends-free with free ends."""
    x, o, e = pen
    base = dict(mismatch=x, gap_opening=o, gap_extension=e)
    for i, kw in enumerate([dict(span="end-to-end", scope="full", heuristic="adaptive"), dict(scope="full", memory_mode="medium"),
                            dict(span="end-to-end", scope="score", heuristic="adaptive"),
                            dict(scope="full", span="ends-free", pattern_begin_free=20, text_end_free=15, heuristic="adaptive", memory_mode="low")]):
        for j, (n, L, err) in enumerate([(300, 1500, 0.05), (100, 3000, 0.08), (1500, 200, 0.1)]):
            batch = datagen.generate(n, L, err, 9500 + 10 * i + j)
            kw2 = common.clamp_free(dict(base, **kw), batch)
            oc, nc = common.configs_pair(**kw2)
            full = oc.scope == 1
            o_ = loader.run(loader.oracle(), oc, batch, want_cigar=full)
            score, status, cigars = common.gpu_run(nc, batch, full, (i + j) % 2 == 0)
            common.assert_same(o_, score, status, cigars, batch, f"banded {kw2} L={L}")


@pytest.mark.parametrize("scope", ["score", "full"])
@pytest.mark.parametrize("order", ["divergent_first", "similar_first"])
def test_pilot_choice_never_changes_results(gpu, scope, order):
    """Batches of >= 64 k pairs let a pilot on the first 8192 pairs pick the first segment width; a batch whose head is
    unlike its tail (very divergent head, similar tail, and the reverse) must still come out right, on a first run and
    on a re-run of the resident batch (which reuses the choice)."""
    head = datagen.generate(9000, 150, 0.14, 9901)
    tail = datagen.generate(61000, 150, 0.01, 9902)
    parts = [head, tail] if order == "divergent_first" else [tail, head]
    # concatenate the two generated batches without going through Python strings
    import numpy as np
    seqs = np.concatenate([parts[0]["seqs"], parts[1]["seqs"]])
    shift = len(parts[0]["seqs"])
    batch = {"seqs": seqs,
             "p_off": np.concatenate([parts[0]["p_off"], parts[1]["p_off"] + shift]),
             "p_len": np.concatenate([parts[0]["p_len"], parts[1]["p_len"]]),
             "t_off": np.concatenate([parts[0]["t_off"], parts[1]["t_off"] + shift]),
             "t_len": np.concatenate([parts[0]["t_len"], parts[1]["t_len"]])}
    oc, nc = common.configs_pair(span="end-to-end", scope=scope)
    full = scope == "full"
    o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
    from pywfa_amd import _native
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    for _ in range(2):
        rb.run(); rb.sync()
        score, status, cig = rb.results(full)
        assert np.array_equal(status, o["status"]) and np.array_equal(score, o["score"])
        if full:
            ops, cb, cl = cig
            for i in range(0, len(score), 37):
                assert ops[cb[i]:cb[i] + cl[i]].tobytes() == o["cigars"][i]
    rb.close(); al.close()


@pytest.mark.parametrize("kw", [dict(span="end-to-end", scope="score"), dict(scope="full"),
                                dict(scope="full", span="ends-free", pattern_begin_free=6, pattern_end_free=9, text_begin_free=4, text_end_free=3),
                                dict(scope="full", heuristic="adaptive"), dict(scope="score", heuristic="X-drop", xdrop=30),
                                dict(scope="full", distance="affine2p"), dict(scope="full", max_steps=14),
                                dict(scope="full", mismatch=4, gap_opening=4, gap_extension=2),
                                dict(scope="full", mismatch=7, gap_opening=3, gap_extension=2),      # no banded shape: the general kernel
                                dict(scope="full", distance="levenshtein")])
def test_single_calls_match_oracle(gpu, kw, monkeypatch):
    """Calls of 1 .. 4 096 short pairs (pywfa's usual loop, and small batches) take the single-call path: one launch of the banded kernel reading the
    host-packed pairs from the pinned block, completion polled by the host; pairs it cannot hold (unrelated sequences: the
    wavefront outgrows 128 diagonals), letters outside ACGT and penalty shapes without a banded instantiation go through the
    general kernel instead.  Every result against the oracle; the same with the polling and the banded form switched off."""
    from pywfa_amd import _native
    rng = np.random.default_rng(4)
    base = datagen.generate(64, 150, 0.04, 4711)
    pats, txts = [], []
    for i in range(64):
        p, t = datagen.pair_strings(base, i)
        k = i % 8
        if k == 1: p = p[:int(rng.integers(0, 150))]
        if k == 2: t = "".join(rng.choice(list("ACGT"), size=300))          # unrelated: handed on
        if k == 3: t = t[:50] + "N" + t[51:]
        if k == 4: p, t = "", t[:9]
        if k == 5: t = t[:60] + t[90:]                                       # a 30-base gap
        pats.append(p); txts.append(t)
    batch = datagen.from_strings(pats, txts)
    if any(kw.get(f, 0) for f in ("pattern_begin_free", "pattern_end_free", "text_begin_free", "text_end_free")):
        batch = datagen.subset(batch, np.flatnonzero((batch["p_len"] >= 10) & (batch["t_len"] >= 10)))   # (free ends fit every pair)
    nb = len(batch["p_len"])
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
    for env in ({}, {"WFA_HIP_NO_TINY_POLL": "1"}, {"WFA_HIP_NO_TINY_BAND": "1"}):
        for k_, v_ in env.items(): monkeypatch.setenv(k_, v_)
        al = _native.Aligner(nc)
        lo = 0
        for size in [1, 1, 2, 3, 7, 16, 1, 17, 64, 300, 1024, 5, 4096, 4097]:   # (17 ..: the banded form only; beyond 4 096 pairs / the pinned block: the batch path)
            idx = np.arange(lo, lo + size) % nb
            sub = datagen.subset(batch, idx)
            score, status, cig = al.align_batch(sub, full)
            assert np.array_equal(score, o["score"][idx]) and np.array_equal(status, o["status"][idx]), (env, size, lo)
            if full:
                ops, cbeg, clen = cig
                for j, i in enumerate(idx):
                    assert ops[cbeg[j]:cbeg[j] + clen[j]].tobytes() == o["cigars"][i], (env, size, lo, j)
            lo += size
        al.close()
        for k_ in env: monkeypatch.delenv(k_)


@pytest.mark.parametrize("kw", [dict(scope="score", span="end-to-end"), dict(scope="full"), dict(scope="full", mismatch=5),
                                dict(scope="full", distance="affine2p"), dict(scope="score", distance="levenshtein"), dict(scope="full", match=-1)])
def test_align_pair_entry_matches_the_batch_entry_and_the_oracle(gpu, kw):
    """wfa_hip_align_pair (round 4: pywfa's one-pair-per-call pattern without arrays on the way) against the oracle: ragged pairs,
    an empty pattern, a letter outside ACGT, sequences in either order in memory."""
    from pywfa_amd import _native
    base = datagen.generate(40, 150, 0.04, 4712)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    al = _native.Aligner(nc)
    try:
        for i in range(40):
            p, t = datagen.pair_strings(base, i)
            if i % 8 == 1: p = p[:i]
            if i % 8 == 3: t = t[:50] + "N" + t[51:]
            if i % 8 == 4: p = ""
            one = datagen.from_strings([p], [t])
            o = loader.run(loader.oracle(), oc, one, want_cigar=full)
            score, status, ops = al.align_pair(p.encode(), t.encode(), full)
            assert (score, status) == (int(o["score"][0]), int(o["status"][0])), (kw, i)
            if full:
                assert ops == o["cigars"][0], (kw, i)
    finally:
        al.close()
    import pywfa_amd
    a = pywfa_amd.WavefrontAligner("TCTTTACTCGCGCGTTGGAGAAATACAATAGT", **{k: v for k, v in kw.items() if k in ("scope", "span")})
    s = a.wavefront_align("TCTATACTGCGCGTTTGGAGAAATAAAATAGT")
    assert s == -24 and (a.cigarstring == "3M1X4M1D7M1I9M1X6M" if kw.get("scope", "full") == "full" else a.cigarstring == "")


WILD = [dict(), dict(scope="score", span="end-to-end"), dict(heuristic="adaptive", span="end-to-end"), dict(distance="affine2p"),
        dict(distance="levenshtein"), dict(span="ends-free", pattern_begin_free=5, text_end_free=9, scope="score"),
        dict(memory_mode="biwfa", span="end-to-end"), dict(wildcard="A", scope="score")]


@pytest.mark.parametrize("cfg_idx", range(len(WILD)))
def test_wildcard_batches_split_by_letters(gpu, cfg_idx):
    """Round 5: with a wildcard letter outside ACGT, pairs whose letters are all ACGT take the 2-bit kernels (nothing in them can match by
    wildcard) and only the pairs holding other letters are aligned on their bytes with the wildcard rule (align.pyx:297-304,
    R/wavefront_extend_kernels.c:142-163).  Mixed batches — clean pairs, pairs with N in the pattern, the text, both, runs of N — against
    the oracle; a wildcard that is one of ACGT keeps every pair on its bytes."""
    rng = np.random.default_rng(400 + cfg_idx)
    pats, txts = [], []
    for b in (datagen.generate(700, 150, 0.03, 71 + cfg_idx), datagen.generate(40, 900, 0.06, 72 + cfg_idx)):
        for i in range(len(b["p_len"])):
            p, t = datagen.pair_strings(b, i)
            u = rng.random()
            if u < 0.25:
                k = int(rng.integers(0, len(p))); p = p[:k] + "N" * int(rng.integers(1, 4)) + p[k + 1:]
            elif u < 0.4:
                k = int(rng.integers(0, len(t))); t = t[:k] + "N" + t[k + 1:]
            elif u < 0.5:
                p = p.replace("ACG", "ANG", 2); t = t.replace("TT", "TN", 1)
            pats.append(p); txts.append(t)
    batch = datagen.from_strings(pats, txts)
    kw = dict(dict(scope="full", wildcard="N"), **WILD[cfg_idx])
    kw = common.clamp_free(kw, batch)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
    for resident in (True, False):
        score, status, cigars = common.gpu_run(nc, batch, full, resident=resident)
        common.assert_same(o, score, status, cigars, batch, f"wildcard {kw} resident={resident}")


LIN = [dict(distance="levenshtein"), dict(distance="levenshtein", span="end-to-end"), dict(distance="linear"),
       dict(distance="linear", mismatch=3, gap_extension=5, span="end-to-end"), dict(distance="linear", mismatch=2, gap_extension=1),
       dict(distance="linear", mismatch=6, gap_extension=2, memory_mode="medium"), dict(distance="indel"), dict(distance="indel", span="end-to-end")]


@pytest.mark.parametrize("env", [{}, {"WFA_HIP_NO_LIN": "1"}, {"WFA_HIP_RTC_FAIL": "1"}])
@pytest.mark.parametrize("cfg_idx", range(len(LIN)))
def test_one_component_distances_with_cigars_on_the_register_kernels(gpu, cfg_idx, env, monkeypatch):
    # (the suite's time budget: the mapping switched off and the failing run-time compiler on two configurations each; WFA_TEST_FULL=1: all)
    if os.environ.get("WFA_TEST_FULL") != "1" and cfg_idx not in ((0, 2, 3, 6) if not env else (0,) if "WFA_HIP_NO_LIN" in env else (2,)):
        pytest.skip("sampled on the suite's time budget (WFA_TEST_FULL=1 runs every cell)")
    """Round 6 (VERDICT r05 missing 1): gap-linear, levenshtein and indel with CIGARs take the lane / segment kernels' LIN form (gap-affine with
    o = 0 and no extension candidates: R/wavefront_compute_linear.c:44-74, R/wavefront_compute_edit.c:44-100 and the linear backtrace's
    choices, R/wavefront_backtrace.c:223-319); what they hand on goes to the general kernel under the original configuration.  Op
    strings against the oracle on reads with many ties (tandem repeats, low complexity), high divergence (pairs the 16-diagonal band
    hands on), letters outside ACGT; the same with the mapping off and with a run-time compiler that fails (general kernel only)."""
    import validate_oracle as vo
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    batches = [datagen.generate(3000, 150, 0.04, 8800 + cfg_idx), datagen.generate(600, 150, 0.15, 8900 + cfg_idx), vo.corpus_special(seed=123),
               datagen.generate(80000 if not env else 20000, 100, 0.02, 9000 + cfg_idx)]
    for bi, batch in enumerate(batches):
        oc, nc = common.configs_pair(**dict(LIN[cfg_idx], scope="full"))
        if bi == 3:   # (the large batch: every 50th pair against the oracle, all of them for completion)
            idx = np.arange(0, len(batch["p_len"]), 50)
            o = loader.run(loader.oracle(), oc, datagen.subset(batch, idx))
            score, status, cigars = common.gpu_run(nc, batch, True, resident=False)
            assert np.array_equal(score[idx], o["score"]) and np.array_equal(status[idx], o["status"]) and int((status != 0).sum()) == 0
            assert [cigars[i] for i in idx] == o["cigars"]
            continue
        o = loader.run(loader.oracle(), oc, batch)
        score, status, cigars = common.gpu_run(nc, batch, True, resident=(bi % 2 == 0))
        common.assert_same(o, score, status, cigars, batch, f"lin {LIN[cfg_idx]} batch {bi} env {env}")
