#!/usr/bin/env python3
"""Fast-kernel parity + timing sweep (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import loader
from pywfa_amd import datagen, _native
sys.path.insert(0, os.path.join(ROOT, "tools"))
import common

def run(batch, kw, label):
    oc, nc = common.configs_pair(**kw)
    t0 = time.time(); o = loader.run(loader.reference() if loader.have_reference() else loader.oracle(), oc, batch, want_cigar=False); t1 = time.time()
    al = _native.Aligner(nc); rb = al.batch(batch)
    rb.run(); rb.sync()
    for _ in range(3): rb.run()
    rb.sync()
    ms, pairs = rb.last_kernel()
    score, status, _ = rb.results(False)
    fb = rb.fallback_pairs(); rb.close(); al.close()
    bad = int(((score != o["score"]) | (status != o["status"])).sum())
    n = len(score)
    print(f"{'OK ' if bad == 0 else 'BAD'} {label:28s} n={n:8d} mism={bad} fallback={fb} ({100.0*fb/max(n,1):.2f}%) kernel_ms={ms:.3f} -> {n/ms/1e3:.1f} M aln/s  cpu={n/(t1-t0)/1e6:.2f} M/s", flush=True)
    if bad:
        i = int(np.flatnonzero((score != o["score"]) | (status != o["status"]))[0])
        print("  first bad", i, o["score"][i], o["status"][i], score[i], status[i], datagen.pair_strings(batch, i))
    return bad

bad = 0
n = int(os.environ.get("N", "1000000"))
for kw, lab in ((dict(span="end-to-end", scope="score"), "e2e"), (dict(scope="score"), "endsfree0")):
    for L, e in ((150, 0.02), (150, 0.05), (150, 0.10), (100, 0.02), (250, 0.02), (50, 0.3), (400, 0.01)):
        bad += run(datagen.generate(n if L <= 150 else n // 4, L, e, 77 + L), kw, f"{lab} L{L} e{e}")
import validate_oracle as vo
bad += run(vo.corpus_special(), dict(span="end-to-end", scope="score"), "special")
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
