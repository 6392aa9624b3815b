"""BASELINE.json configurations C4 and C5 as they are written, through the C ABI, against the oracle.

C4 = 10 kb pairs (seed 1004, 8 %), distance=affine2p, span=ends-free with 100 free bases on both pattern ends,
     the text cut by 50 bases at both ends, full CIGAR, **no heuristic**.
C5 = 100 kb pairs (seed 1005, 8 %), X-drop.  With pywfa's match=0 the reference's X-drop score falls with progress
     (R/wavefront_heuristic.c:306-307), so X-drop(20) drops every pair after a few steps
     (status 1, score INT32_MIN, empty CIGAR); a very large xdrop or match<0 lets pairs complete.  wf-adaptive at
     100 kb exercises the 32-bit history entries of the banded kernel (sequences >= 32 000 bases).
"""
import os

import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import _native, datagen

pytestmark = pytest.mark.gpu

C4_KW = dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full")
INT32_MIN = -2147483648


def c4_batch(n):
    return datagen.trim_text(datagen.generate(n, 10000, 0.08, datagen.SEEDS["C4"]), 50)


def test_c4_as_written_exact_full_cigar(gpu):
    """No heuristic: 2p wavefronts grow to ~9 k diagonals and the explicit history to ~0.4 GB per pair; the pairs must
    come back COMPLETED (never -200) with the reference's op strings."""
    batch = c4_batch(8)
    oc, nc = common.configs_pair(**C4_KW)
    o = loader.run(loader.oracle(), oc, batch)
    for resident in (True, False):
        score, status, cigars = common.gpu_run(nc, batch, True, resident)
        assert (status == 0).all(), status
        common.assert_same(o, score, status, cigars, batch, f"C4 exact resident={resident}")


def test_c4_as_written_64_pairs_vs_the_real_library(gpu):
    """the newest path (gap-affine-2p wide kernel: workspace rows, seven-bit origin codes, five-component
    walk) on 64 pairs of C4 as BASELINE writes it, against the real WFA2-lib run on every host thread (scores, statuses and op
    strings); falls back on the oracle where the reference build did not travel."""
    batch = c4_batch(64)
    oc, nc = common.configs_pair(**C4_KW)
    if loader.have_reference():
        o = loader.reference_mt_full(oc, batch)
    else:
        o = loader.run(loader.oracle(), oc, batch)
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    rb.run(); rb.sync()
    score, status, (ops, cbeg, clen) = rb.results(True)
    assert rb.fallback_pairs() == 0
    rb.close(); al.close()
    cigars = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in range(len(score))]
    common.assert_same(o, score, status, cigars, batch, "C4 as written, 64 pairs")
    assert (status == 0).all()


@pytest.mark.parametrize("memory_mode", ["medium", "low"])
def test_c4_exact_low_memory_modes(gpu, memory_mode):
    """memory_mode medium / low (piggy-back history) returns the same alignments for gap-affine-2p."""
    batch = c4_batch(6)
    oc, nc = common.configs_pair(**dict(C4_KW, memory_mode=memory_mode))
    o = loader.run(loader.oracle(), oc, batch)
    score, status, cigars = common.gpu_run(nc, batch, True, True)
    common.assert_same(o, score, status, cigars, batch, f"C4 exact {memory_mode}")


@pytest.mark.parametrize("memory_mode", ["high", "medium"])
def test_c4_adaptive_sample_vs_oracle(gpu, memory_mode):
    batch = c4_batch(96)
    kw = dict(C4_KW, heuristic="adaptive", memory_mode=memory_mode)
    oc, nc = common.configs_pair(**kw)
    o = loader.run(loader.oracle(), oc, batch)
    score, status, cigars = common.gpu_run(nc, batch, True, True)
    common.assert_same(o, score, status, cigars, batch, f"C4 adaptive {memory_mode}")


def test_c4_full_size_adaptive_properties(gpu):
    """C4 at BASELINE's size (1 M x 10 kb; WFA_C4_PAIRS shrinks it) with wf-adaptive — stated: the exact form moves
    ~0.4 GB of history per pair — through properties: every transcript valid for its (trimmed) sequences, a seeded
    sample equal to the oracle, and idempotent across two runs."""
    n = int(os.environ.get("WFA_C4_PAIRS", "1000000"))
    batch = c4_batch(n)
    kw = dict(C4_KW, heuristic="adaptive")
    oc, nc = common.configs_pair(**kw)
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    rb.run(); rb.sync()
    score, status, (ops, cbeg, clen) = rb.results(True)
    s1 = score.copy()
    rb.run(); rb.sync()
    score2, status2, _ = rb.results(False)
    rb.close(); al.close()
    assert np.array_equal(s1, score2)
    assert (status == 0).all()
    # ends-free: the transcript covers both sequences completely (free ends appear as leading / trailing I / D), so the
    # validity walk applies; the score of an ends-free alignment excludes the free ends, hence no penalty check here
    bad, first = loader.check_cigars(oc, batch, score, ops, cbeg, clen, check_score=False)
    assert bad == 0, f"{bad} invalid transcripts, first at pair {first}"
    idx = np.arange(0, n, max(1, n // 48))
    sub = datagen.subset(batch, idx)
    o = loader.run(loader.oracle(), oc, sub)
    cigs = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in idx]
    common.assert_same(o, score[idx], status[idx], cigs, sub, "C4 sample vs oracle")


# ---------------------------------------------------------------------------------------------- C5
def c5_batch(n):
    return datagen.generate(n, 100000, 0.08, datagen.SEEDS["C5"])


@pytest.mark.parametrize("scope", ["score", "full"])
def test_c5_xdrop20_drops_every_pair(gpu, scope):
    batch = c5_batch(24)
    oc, nc = common.configs_pair(span="end-to-end", scope=scope, heuristic="X-drop", xdrop=20)
    o = loader.run(loader.oracle(), oc, batch)
    score, status, cigars = common.gpu_run(nc, batch, scope == "full", True)
    common.assert_same(o, score, status, cigars, batch, f"C5 xdrop 20 {scope}")
    assert (status == 1).all()
    if scope == "full":
        assert (score == INT32_MIN).all() and all(len(c) == 0 for c in cigars)


def test_c5_xdrop_large_enough_to_complete(gpu):
    """match=0: the X-drop score at the end is -(plen + tlen + s) / 2, so only an xdrop beyond that keeps the optimum;
    match=-1: the score grows with progress and xdrop=100 completes."""
    for n, kw in ((4, dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=400000)),
                  (16, dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=100, match=-1))):
        batch = c5_batch(n)
        oc, nc = common.configs_pair(**kw)
        # (an exact 100 kb run is ~2 s per pair in WFA2-lib and ~10 s in the plain restatement: the real library when it is there)
        checker = loader.reference() if (n == 4 and loader.have_reference()) else loader.oracle()
        o = loader.run(checker, oc, batch)
        score, status, _ = common.gpu_run(nc, batch, False, True)
        common.assert_same(o, score, status, None, batch, f"C5 {kw}")
        assert (status == 0).all()
    # full CIGAR with match<0 on a shorter prefix of the same pairs (the explicit history of a 100 kb exact run is ~6 GB)
    sub = {"seqs": batch["seqs"], "p_off": batch["p_off"], "p_len": np.minimum(batch["p_len"], 20000).astype(np.int32),
           "t_off": batch["t_off"], "t_len": np.minimum(batch["t_len"], 20000).astype(np.int32)}
    kw = dict(span="end-to-end", scope="full", heuristic="X-drop", xdrop=100, match=-1)
    oc, nc = common.configs_pair(**kw)
    o = loader.run(loader.oracle(), oc, sub)
    score, status, cigars = common.gpu_run(nc, sub, True, True)
    common.assert_same(o, score, status, cigars, sub, f"C5 20 kb prefix {kw}")


@pytest.mark.parametrize("kw", [dict(span="end-to-end", scope="full", heuristic="adaptive"),
                                dict(span="end-to-end", scope="full", heuristic="adaptive", memory_mode="medium"),
                                dict(distance="affine2p", scope="score", heuristic="adaptive")])
def test_c5_adaptive_100kb_vs_oracle(gpu, kw):
    """Sequences >= 32 000 bases: the banded kernel's history entries are 4 x int32 (h16 = 0)."""
    batch = c5_batch(16)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
    for resident in (True, False):
        score, status, cigars = common.gpu_run(nc, batch, full, resident)
        common.assert_same(o, score, status, cigars, batch, f"C5 adaptive {kw}")


def test_c5_thousand_pairs_properties(gpu):
    """A 1 000-pair prefix of C5 (200 MB of sequence): X-drop(20) drops all of it; wf-adaptive gives valid transcripts whose
    gap-affine penalty is -score for every pair, the same scores with scope=score, and the oracle's op strings on a sample."""
    n = int(os.environ.get("WFA_C5_PAIRS", "1000"))
    batch = c5_batch(n)
    _, nc = common.configs_pair(span="end-to-end", scope="full", heuristic="X-drop", xdrop=20)
    score, status, cigars = common.gpu_run(nc, batch, True, True)
    assert (status == 1).all() and (score == INT32_MIN).all() and all(len(c) == 0 for c in cigars)
    oc, nc = common.configs_pair(span="end-to-end", scope="full", heuristic="adaptive")
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    rb.run(); rb.sync()
    score, status, (ops, cbeg, clen) = rb.results(True)
    rb.close(); al.close()
    assert (status == 0).all()
    bad, first = loader.check_cigars(oc, batch, score, ops, cbeg, clen, check_score=True)
    assert bad == 0, f"{bad} invalid transcripts, first at pair {first}"
    _, nc_s = common.configs_pair(span="end-to-end", scope="score", heuristic="adaptive")
    score_s, status_s, _ = common.gpu_run(nc_s, batch, False, True)
    assert np.array_equal(score_s, score) and (status_s == 0).all()
    idx = np.arange(0, n, max(1, n // 12))
    sub = datagen.subset(batch, idx)
    o = loader.run(loader.oracle(), oc, sub)
    cigs = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in idx]
    common.assert_same(o, score[idx], status[idx], cigs, sub, "C5 adaptive sample vs oracle")


# ------------------------------------------------------------------------ handles
def test_config_change_after_batch_create_does_not_reach_the_batch(gpu):
    """A resident batch is laid out under the configuration in force when it was created (op regions, work lists, checked
    free ends): it keeps running under that configuration whatever set_config does afterwards."""
    batch = datagen.generate(3000, 150, 0.03, 4242)
    oc_s, nc_s = common.configs_pair(span="end-to-end", scope="score")
    oc_f, nc_f = common.configs_pair(span="ends-free", scope="full", pattern_begin_free=10, text_end_free=10, wildcard="N")
    o_s = loader.run(loader.oracle(), oc_s, batch)
    al = _native.Aligner(nc_s)
    rb = al.batch(batch)
    al.set_config(nc_f)            # scope score -> full, free ends, wildcard
    rb.run(); rb.sync()
    score, status, _ = rb.results(False)
    assert np.array_equal(score, o_s["score"]) and np.array_equal(status, o_s["status"])
    # and a batch created now runs under the new configuration
    o_f = loader.run(loader.oracle(), oc_f, batch)
    rb2 = al.batch(batch)
    rb2.run(); rb2.sync()
    s2, st2, (ops, cb, cl) = rb2.results(True)
    common.assert_same(o_f, s2, st2, [ops[cb[i]:cb[i] + cl[i]].tobytes() for i in range(len(s2))], batch, "after set_config")
    rb.close(); rb2.close(); al.close()


def test_batch_outlives_aligner_handle(gpu):
    """wfa_hip_destroy with resident batches alive only marks the handle; the last batch frees it (through the raw C ABI:
    the Python wrapper closes its batches first)."""
    import ctypes
    L = _native.lib()
    batch = datagen.generate(2000, 150, 0.02, 777)
    oc, nc = common.configs_pair(span="end-to-end", scope="score")
    o = loader.run(loader.oracle(), oc, batch)
    seqs, p_off, p_len, t_off, t_len, n = _native._check_batch(batch)
    h = L.wfa_hip_create(ctypes.byref(nc), 0)
    assert h
    b = L.wfa_hip_batch_create(h, n, _native._ptr(seqs), _native._ptr(p_off), _native._ptr(p_len), _native._ptr(t_off), _native._ptr(t_len))
    assert b
    L.wfa_hip_destroy(h)                      # handle marked, batch still valid
    assert L.wfa_hip_batch_run(b, None) == 0
    score = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
    assert L.wfa_hip_batch_results(b, _native._ptr(score), _native._ptr(status), None, None, None, None) == 0
    L.wfa_hip_batch_destroy(b)                # frees the aligner too
    assert np.array_equal(score, o["score"]) and np.array_equal(status, o["status"])
    # the Python wrapper: closing the aligner closes its batches, closing them again is harmless
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    al.close()
    rb.close()


def test_runs_on_different_streams_are_ordered(gpu):
    """Two resident batches of one aligner share its workspace; runs enqueued on different streams are ordered by the
    library (full-CIGAR runs of 1.5 kb reads keep their history in that workspace)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    streams = []
    for _ in range(2):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(s)) == 0
        streams.append(s)
    b1 = datagen.generate(3000, 1500, 0.06, 31337)
    b2 = datagen.generate(3000, 1500, 0.06, 31338)
    oc, nc = common.configs_pair(span="end-to-end", scope="full", heuristic="adaptive")
    o1 = loader.run(loader.oracle(), oc, b1)
    o2 = loader.run(loader.oracle(), oc, b2)
    al = _native.Aligner(nc)
    r1, r2 = al.batch(b1), al.batch(b2)
    for _ in range(3):
        r1.run(streams[0]); r2.run(streams[1]); r1.run(streams[1]); r2.run(streams[0])
    for rb, o, bt in ((r1, o1, b1), (r2, o2, b2)):
        rb.sync()
        s, st, (ops, cb, cl) = rb.results(True)
        common.assert_same(o, s, st, [ops[cb[i]:cb[i] + cl[i]].tobytes() for i in range(len(s))], bt, "two streams")
    r1.close(); r2.close(); al.close()
    for s in streams:
        hip.hipStreamDestroy(s)


# ------------------------------------------------------------------------ several devices
@pytest.mark.parametrize("scope", ["score", "full"])
def test_multi_device_entry_shards_and_merges(gpu, scope):
    """wfa_hip_multi_align_batch: contiguous shards, one host thread + aligner per entry of devices[], results in disjoint
    slices of the caller's arrays.  On a one-GPU box the entries are [0, 0, 0] (three threads feeding one device: the
    sharding, threading and merging are the same code); with two or more GPUs the shards really run on different devices."""
    ndev = gpu
    devices = list(range(ndev)) if ndev >= 2 else [0, 0, 0]
    rng = np.random.default_rng(8)
    pats, txts = [], []
    for i in range(5000):
        L = int(rng.choice([30, 150, 150, 400, 1200]))
        b1 = datagen.generate(1, L, float(rng.choice([0.0, 0.03, 0.1])), 50_000 + i)
        p_, t_ = datagen.pair_strings(b1, 0)
        pats.append(p_); txts.append(t_)
    batch = datagen.from_strings(pats, txts)
    oc, nc = common.configs_pair(scope=scope)
    o = loader.run(loader.oracle(), oc, batch)
    m = _native.MultiAligner(nc, devices)
    for _ in range(2):
        score, status, cig = m.align_batch(batch, scope == "full")
        cigars = None
        if cig is not None:
            ops, cbeg, clen = cig
            cigars = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in range(len(score))]
        common.assert_same(o, score, status, cigars, batch, f"multi {devices} {scope}")
    m.close()
    # the Python class: devices=[...] routes batches through the same entry
    import pywfa_amd
    a = pywfa_amd.WavefrontAligner(scope=scope, devices=devices)
    out = a.wavefront_align_batch(txts[:500], pats[:500])
    assert out["score"].tolist() == o["score"][:500].tolist()
    a.close()


def test_two_devices_parity(gpu):
    if gpu < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); the one-GPU box runs test_multi_device_entry_shards_and_merges")
    batch = datagen.generate(200000, 150, 0.02, 606)
    oc, nc = common.configs_pair(span="end-to-end", scope="score")
    m = _native.MultiAligner(nc, [0, 1])
    score, status, _ = m.align_batch(batch, False)
    m.close()
    s1, st1, _ = common.gpu_run(nc, batch, False, False)
    assert np.array_equal(score, s1) and np.array_equal(status, st1)


@pytest.mark.parametrize("scope", ["score", "full"])
def test_host_packed_upload_matches_the_device_packed_one(gpu, scope, monkeypatch):
    """Batches of >= 256 k pairs are packed to 2 bits per base by host threads on their way into the pinned upload ring
    (csrc/host_pack.cpp); WFA_HIP_HOST_PACK=0 keeps the ASCII upload + device pack kernel.  Both give the same results, on
    a ragged batch with letters outside ACGT sprinkled in (those pairs are aligned on their bytes), and a sample agrees with
    the oracle."""
    rng = np.random.default_rng(77)
    base = datagen.generate(300000, 150, 0.03, 4242)
    pats, txts = [], []
    for i in range(2000):   # ragged head: other lengths, N / lower case / empty
        p, t = datagen.pair_strings(base, i)
        k = i % 7
        if k == 0: p = p[:int(rng.integers(0, 150))]
        if k == 1: t = t[:80] + "N" + t[81:]
        if k == 2: p = p.lower()
        if k == 3: p, t = "", t[:33]
        if k == 4: p = p * 3
        pats.append(p); txts.append(t)
    head = datagen.from_strings(pats, txts, upper=False)
    shift = len(head["seqs"])
    batch = dict(seqs=np.concatenate([head["seqs"], base["seqs"]]),
                 p_off=np.concatenate([head["p_off"], base["p_off"][2000:] + shift]), p_len=np.concatenate([head["p_len"], base["p_len"][2000:]]),
                 t_off=np.concatenate([head["t_off"], base["t_off"][2000:] + shift]), t_len=np.concatenate([head["t_len"], base["t_len"][2000:]]))
    oc, nc = common.configs_pair(span="end-to-end", scope=scope)
    full = scope == "full"
    res = {}
    for hp in ("1", "0"):
        monkeypatch.setenv("WFA_HIP_HOST_PACK", hp)
        al = _native.Aligner(nc)     # (the knobs are read when the aligner is created)
        res[hp] = al.align_batch(batch, full)
        al.close()
    s1, st1, c1 = res["1"]; s0, st0, c0 = res["0"]
    assert np.array_equal(s1, s0) and np.array_equal(st1, st0)
    if full:
        assert np.array_equal(c1[1], c0[1]) and np.array_equal(c1[2], c0[2])
        sel = np.r_[0:2500, 299000:300000]
        for i in sel:
            assert c1[0][c1[1][i]:c1[1][i] + c1[2][i]].tobytes() == c0[0][c0[1][i]:c0[1][i] + c0[2][i]].tobytes()
    sel = np.r_[0:2500, 299000:300000]
    sub = datagen.subset(batch, sel)
    o = loader.run(loader.oracle(), oc, sub, want_cigar=full)
    assert np.array_equal(s1[sel], o["score"]) and np.array_equal(st1[sel], o["status"])
    if full:
        for j, i in enumerate(sel):
            assert c1[0][c1[1][i]:c1[1][i] + c1[2][i]].tobytes() == o["cigars"][j]


def test_large_batch_runs_on_a_caller_stream_right_after_create(gpu):
    """Batches of >= 256 k pairs come back from wfa_hip_batch_create with the DMAs of the host-packed upload still in
    flight on the library's stream; a run on a caller-created stream is ordered after them by the library."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0   # hipStreamNonBlocking
    batch = datagen.generate(400000, 150, 0.02, 919)
    oc, nc = common.configs_pair(span="end-to-end", scope="score")
    o = loader.run(loader.reference() if loader.have_reference() else loader.oracle(), oc, batch, want_cigar=False)
    al = _native.Aligner(nc)
    for _ in range(3):
        rb = al.batch(batch)
        rb.run(s)
        rb.sync()
        score, status, _ = rb.results(False)
        rb.close()
        assert np.array_equal(score, o["score"]) and np.array_equal(status, o["status"])
    al.close()
    hip.hipStreamDestroy(s)


@pytest.mark.parametrize("n,length,kw", [(3000, 150, dict(span="end-to-end", scope="score")), (3000, 150, dict(scope="full")),
                                         (300000, 150, dict(span="end-to-end", scope="score")),
                                         (40, 3000, dict(scope="full", heuristic="adaptive")),
                                         (5, 100, dict(distance="affine2p", scope="full"))])
def test_packed2bits_entry_equals_the_ascii_entry(gpu, n, length, kw):
    """wfa_hip_align_batch_packed2bits / wfa_hip_batch_create_packed2bits (cf. wavefront_align_packed2bits, wfa.h:211): 2-bit
    reads in the reference's packed form give the results of the ASCII entry on the decoded sequences, and those of the oracle."""
    batch = datagen.generate(n, length, 0.04, 5150 + n)
    if n == 3000:   # ragged: other lengths, empty sequences
        pats, txts = [], []
        for i in range(n):
            p, t = datagen.pair_strings(batch, i)
            if i % 11 == 0: p = p[:i % 150]
            if i % 13 == 0: t = ""
            pats.append(p); txts.append(t)
        batch = datagen.from_strings(pats, txts)
    pk = datagen.to_packed2bits(batch)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    o = loader.run(loader.oracle(), oc, datagen.subset(batch, np.arange(min(n, 3000))), want_cigar=full)
    al = _native.Aligner(nc)
    ref = al.align_batch(batch, full)
    got = al.align_batch(pk, full)
    rb = al.batch(pk); rb.run(); rb.sync()
    res = rb.results(full)
    rb.close(); al.close()
    for s, st, cig in (got, res):
        assert np.array_equal(s, ref[0]) and np.array_equal(st, ref[1])
        m = min(n, 3000)
        cigars = None
        if full:
            ops, cb, cl = cig
            cigars = [ops[cb[i]:cb[i] + cl[i]].tobytes() for i in range(m)]
            rops, rcb, rcl = ref[2]
            assert all(cigars[i] == rops[rcb[i]:rcb[i] + rcl[i]].tobytes() for i in range(m))
        common.assert_same(o, s[:m], st[:m], cigars, batch, f"packed2bits {kw}")


def test_packed2bits_refuses_a_wildcard(gpu):
    oc, nc = common.configs_pair(wildcard="N")
    al = _native.Aligner(nc)
    with pytest.raises(NotImplementedError):
        al.align_batch(datagen.to_packed2bits(datagen.from_strings(["ACGT"], ["ACGA"])), True)
    al.close()


def test_few_long_reads_take_the_pinned_host_packed_upload(gpu, monkeypatch):
    """Round 4: batches of few pairs but many bases (C5: 200 KB of ASCII per pair) take the pinned, host-packed upload ring too
    (it used to start at 256 k pairs; a plain copy from pageable memory moved C5's reads at 3.6 GB/s).  600 x 30 kb reads, some with
    a letter outside ACGT, wf-adaptive: the same results with the ring (default), without it (WFA_HIP_NO_PIPE=1) and with the
    ASCII ring + device pack (WFA_HIP_HOST_PACK=0); a sample against the oracle; and through the multi-device entry."""
    base = datagen.generate(600, 30000, 0.05, 4343)
    pats, txts = [], []
    for i in range(600):
        p, t = datagen.pair_strings(base, i)
        if i % 97 == 5: t = t[:1000] + "N" + t[1001:]
        if i % 211 == 7: p = p[:12345]
        pats.append(p); txts.append(t)
    batch = datagen.from_strings(pats, txts, upper=False)
    assert int(batch["p_len"].sum() + batch["t_len"].sum()) >= (32 << 20)
    oc, nc = common.configs_pair(span="end-to-end", scope="score", heuristic="adaptive")
    res = {}
    for name, env in (("ring", {}), ("plain", {"WFA_HIP_NO_PIPE": "1"}), ("ascii-ring", {"WFA_HIP_HOST_PACK": "0"})):
        for k_, v_ in env.items(): monkeypatch.setenv(k_, v_)
        al = _native.Aligner(nc)
        res[name] = al.align_batch(batch, False)
        al.close()
        for k_ in env: monkeypatch.delenv(k_)
    for name in ("plain", "ascii-ring"):
        assert np.array_equal(res["ring"][0], res[name][0]) and np.array_equal(res["ring"][1], res[name][1]), name
    sel = np.r_[0:12, 95:110, 205:215]
    o = loader.run(loader.oracle(), oc, datagen.subset(batch, sel), want_cigar=False)
    assert np.array_equal(res["ring"][0][sel], o["score"]) and np.array_equal(res["ring"][1][sel], o["status"])
    ma = _native.MultiAligner(nc, [0])
    s, st, _ = ma.align_batch(batch, False)
    ma.close()
    assert np.array_equal(s, res["ring"][0]) and np.array_equal(st, res["ring"][1])
