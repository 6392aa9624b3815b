"""BASELINE.json's full sizes through size-independent properties (the oracle cannot walk 10 M pairs
in seconds): C2 = 10 M x 150 bp, gap-affine end-to-end.
  * scope=score and scope=full report the same score for every pair;
  * every transcript is valid for its sequences and its gap-affine penalty equals -score
    (oracle/wfa_oracle.c: wfa_oracle_check_cigars, run over ALL pairs);
  * identical sequences score 0; the score is symmetric under swapping pattern and text;
  * two runs give identical results (idempotence);
  * a seeded sample is compared with the oracle bit-for-bit.
Set WFA_FULL_PAIRS to override the pair count (default 10,000,000).
"""
import os

import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import _native, datagen

pytestmark = pytest.mark.gpu

N_FULL = int(os.environ.get("WFA_FULL_PAIRS", "10000000"))


@pytest.fixture(scope="module")
def c2_batch():
    return datagen.generate(N_FULL, 150, 0.02, datagen.SEEDS["C2"])


def test_c2_full_size_properties(gpu, c2_batch):
    batch = c2_batch
    n = len(batch["p_len"])
    oc_s, nc_s = common.configs_pair(span="end-to-end", scope="score")
    oc_f, nc_f = common.configs_pair(span="end-to-end", scope="full")
    al = _native.Aligner(nc_s)
    rb = al.batch(batch)
    rb.run(); rb.sync()
    score1, status1, _ = rb.results(False)
    rb.run(); rb.sync()
    score2, status2, _ = rb.results(False)
    rb.close(); al.close()
    assert np.array_equal(score1, score2) and np.array_equal(status1, status2)   # idempotent
    assert (status1 == 0).all()
    assert score1.max() <= 0 and score1.min() >= -(4 * 150 + 6 + 2 * 150)
    # full scope: same scores, valid transcripts whose penalty is -score (all pairs)
    al = _native.Aligner(nc_f)
    rb = al.batch(batch)
    rb.run(); rb.sync()
    score_f, status_f, (ops, cbeg, clen) = rb.results(True)
    rb.close(); al.close()
    assert np.array_equal(score_f, score1) and (status_f == 0).all()
    bad, first = loader.check_cigars(oc_f, batch, score_f, ops, cbeg, clen, check_score=True)
    assert bad == 0, f"{bad} invalid transcripts, first at pair {first}"
    # bit-exact against the oracle on a seeded sample
    idx = np.random.default_rng(5).choice(n, size=min(n, 200000), replace=False)
    idx.sort()
    sub = datagen.subset(batch, idx)
    o = loader.run(loader.oracle(), oc_f, sub)
    cigs = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in idx]
    common.assert_same(o, score_f[idx], status_f[idx], cigs, sub, "C2 sample vs oracle")


def test_c2_symmetry_and_identity(gpu, c2_batch):
    batch = c2_batch
    n = min(len(batch["p_len"]), 2000000)
    sub = datagen.subset(batch, np.arange(n))
    swapped = {"seqs": sub["seqs"], "p_off": sub["t_off"], "p_len": sub["t_len"], "t_off": sub["p_off"], "t_len": sub["p_len"]}
    ident = {"seqs": sub["seqs"], "p_off": sub["p_off"], "p_len": sub["p_len"], "t_off": sub["p_off"], "t_len": sub["p_len"]}
    _, nc = common.configs_pair(span="end-to-end", scope="score")
    s0, st0, _ = common.gpu_run(nc, sub, False, resident=True)
    s1, st1, _ = common.gpu_run(nc, swapped, False, resident=True)
    s2, st2, _ = common.gpu_run(nc, ident, False, resident=True)
    assert np.array_equal(s0, s1)          # gap-affine with equal I/D penalties is symmetric
    assert (s2 == 0).all() and (st2 == 0).all()


def test_c3_10kb_adaptive_full_properties(gpu):
    """C3 = 1 M x 10 kb ONT-like pairs, ~8 % error, gap-affine + adaptive, full CIGAR (seed 1003; needs ~60 GB
    of host RAM, set WFA_C3_PAIRS to shrink): every transcript valid with penalty == -score, sample equal to
    the oracle."""
    n = int(os.environ.get("WFA_C3_PAIRS", "1000000"))
    batch = datagen.generate(n, 10000, 0.08, datagen.SEEDS["C3"])
    oc, nc = common.configs_pair(span="end-to-end", scope="full", heuristic="adaptive")
    al = _native.Aligner(nc)
    rb = al.batch(batch)
    rb.run(); rb.sync()
    score, status, (ops, cbeg, clen) = rb.results(True)
    rb.close(); al.close()
    assert (status == 0).all()
    bad, first = loader.check_cigars(oc, batch, score, ops, cbeg, clen, check_score=True)
    assert bad == 0, f"{bad} invalid transcripts, first at pair {first}"
    idx = np.arange(0, n, max(1, n // 40))
    sub = datagen.subset(batch, idx)
    o = loader.run(loader.oracle(), oc, sub)
    cigs = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in idx]
    common.assert_same(o, score[idx], status[idx], cigs, sub, "C3 sample vs oracle")
