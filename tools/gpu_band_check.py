#!/usr/bin/env python3
"""Band-kernel parity sweep (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import loader
from pywfa_amd import datagen, _native
import common
import validate_oracle as vo

def run(batch, kw, label):
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    fn = loader.reference() if loader.have_reference() else loader.oracle()
    t0 = time.time(); o = loader.run(fn, oc, batch); t1 = time.time()
    al = _native.Aligner(nc); rb = al.batch(batch)
    rb.run(); rb.sync()
    rb.run(); rb.sync()
    ms, pairs = rb.last_kernel()
    score, status, cig = rb.results(full)
    fb = rb.fallback_pairs(); rb.close(); al.close()
    n = len(score)
    badidx = np.flatnonzero((score != o["score"]) | (status != o["status"]))
    bad = badidx.size
    first = int(badidx[0]) if bad else -1
    if full:
        ops, cbeg, clen = cig
        for i in range(n):
            if ops[cbeg[i]:cbeg[i]+clen[i]].tobytes() != o["cigars"][i]:
                bad += 1
                if first < 0: first = i
    print(f"{'OK ' if bad == 0 else 'BAD'} {label:34s} n={n:7d} mism={bad} fallback={fb} ({100.0*fb/max(n,1):.2f}%) kernel_ms={ms:.3f} -> {n/ms/1e3:.3f} M aln/s  cpu={n/(t1-t0)/1e6:.4f} M/s", flush=True)
    if bad:
        i = first
        print("  first bad", i, o["score"][i], o["status"][i], score[i], status[i])
        if full: print("   exp", common.rle(o["cigars"][i])[:300]); print("   got", common.rle(ops[cbeg[i]:cbeg[i]+clen[i]].tobytes())[:300])
    return bad

bad = 0
N = int(os.environ.get("N", "20000"))
for scope in ("score", "full"):
    for heur in (None, "adaptive"):
        kw = dict(span="end-to-end", scope=scope, heuristic=heur)
        for L, e, n in ((150, 0.02, N), (150, 0.10, N // 2), (300, 0.05, N // 4), (1000, 0.02, N // 10), (1000, 0.08, N // 10), (3000, 0.08, N // 40), (10000, 0.08, N // 100), (10000, 0.03, N // 100)):
            if heur is None and L >= 3000 and scope == "full": n = max(4, n // 10)
            bad += run(datagen.generate(max(n, 4), L, e, 500 + L), kw, f"{scope} {heur} L{L} e{e}")
    bad += run(vo.corpus_special(), dict(span="end-to-end", scope=scope), f"{scope} special")
    bad += run(vo.corpus_special(), dict(scope=scope, heuristic="adaptive"), f"{scope} special adaptive")
for scope in ("score", "full"):
    for heur in (None, "adaptive"):
        for free in ((8, 7, 3, 2), (20, 0, 0, 9), (0, 30, 25, 0)):
            kw = dict(span="ends-free", scope=scope, heuristic=heur, pattern_begin_free=free[0], pattern_end_free=free[1], text_begin_free=free[2], text_end_free=free[3])
            for L, e, n in ((150, 0.05, N // 2), (1000, 0.08, N // 20), (10000, 0.08, N // 200)):
                if heur is None and L >= 3000: continue
                bad += run(datagen.generate(max(n, 4), L, e, 900 + L), kw, f"EF{free} {scope} {heur} L{L}")
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
