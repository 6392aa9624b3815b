"""Penalty shapes without an instantiation in the library (run-time compiled kernels, csrc/wfa_rtc.cpp): parity + rates."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from oracle import loader
from pywfa_amd import datagen, _native

def run(label, n, L, e, kw, cpu_n=20000):
    batch = datagen.generate(n, L, e, 1002)
    kw = common.clamp_free(kw, batch)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    t0 = time.time()
    al = _native.Aligner(nc); rb = al.batch(batch); rb.run(); rb.sync()
    t_first = time.time() - t0
    t0 = time.time()
    for _ in range(3): rb.run()
    rb.sync(); wall = (time.time() - t0) / 3
    score, status, cig = rb.results(full)
    fb = rb.fallback_pairs(); rb.close(); al.close()
    cpu_n = min(n, cpu_n)
    o = loader.run(loader.oracle(), oc, datagen.subset(batch, np.arange(cpu_n)), want_cigar=full)
    bad = int(((score[:cpu_n] != o["score"]) | (status[:cpu_n] != o["status"])).sum())
    if full:
        ops, cbeg, clen = cig
        bad += sum(1 for i in range(cpu_n) if ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() != o["cigars"][i])
    print(f"{label:56s} aln/s={n / wall:.4g} first_run_s={t_first:.2f} mism={bad} to_general={fb}", flush=True)

S = dict(span="end-to-end", scope="score")
for name, pen in [("default 4/6/2", {}), ("mismatch=5 (5/6/2)", dict(mismatch=5)), ("3/5/1", dict(mismatch=3, gap_opening=5, gap_extension=1)),
                  ("2/8/1", dict(mismatch=2, gap_opening=8, gap_extension=1)), ("7/11/3", dict(mismatch=7, gap_opening=11, gap_extension=3))]:
    run(f"150bp 2% score {name}", 2_000_000, 150, 0.02, dict(S, **pen))
    run(f"150bp 2% full ends-free {name}", 1_000_000, 150, 0.02, dict(scope="full", **pen))
run("150bp adaptive score mismatch=5", 1_000_000, 150, 0.02, dict(S, heuristic="adaptive", mismatch=5))
run("1kb 5% full mismatch=5", 100_000, 1000, 0.05, dict(span="end-to-end", scope="full", mismatch=5), cpu_n=2000)
run("10kb adaptive full mismatch=5", 20_000, 10000, 0.08, dict(span="end-to-end", scope="full", heuristic="adaptive", mismatch=5), cpu_n=100)
C4 = dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive")
run("C4-adaptive default 2p", 10_000, 10000, 0.08, dict(C4), cpu_n=50)
run("C4-adaptive 2p mismatch=5", 10_000, 10000, 0.08, dict(C4, mismatch=5), cpu_n=50)
run("150bp 2p mismatch=5 full (tests/test.py:229)", 500_000, 150, 0.02, dict(distance="affine2p", mismatch=5, scope="full"))
# round 4, item 7: configurations mapped to gap-affine with a translated score
run("150bp levenshtein score", 2_000_000, 150, 0.02, dict(distance="levenshtein", span="end-to-end", scope="score"))
run("150bp indel score", 2_000_000, 150, 0.02, dict(distance="indel", span="end-to-end", scope="score"))
run("150bp linear score", 2_000_000, 150, 0.02, dict(distance="linear", span="end-to-end", scope="score"))
run("150bp affine match=-1 score", 2_000_000, 150, 0.02, dict(span="end-to-end", scope="score", match=-1))
run("150bp affine match=-1 full (pywfa default span)", 1_000_000, 150, 0.02, dict(scope="full", match=-1))
run("150bp 2p match=-2 full", 200_000, 150, 0.02, dict(distance="affine2p", scope="full", match=-2))
run("10kb adaptive full match=-1", 10_000, 10000, 0.08, dict(span="end-to-end", scope="full", heuristic="adaptive", match=-1), cpu_n=50)
