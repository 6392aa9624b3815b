#!/usr/bin/env python3
"""Wall time of one run + sync of a resident 150 bp batch by size (development aid): where the fixed cost of a run (its launches) meets the work."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from pywfa_amd import datagen, _native
for scope in ("score", "full"):
    oc, nc = common.configs_pair(span="end-to-end", scope=scope)
    al = _native.Aligner(nc)
    for n in (8192, 16384, 65536, 262144, 1048576):
        batch = datagen.generate(n, 150, 0.02, 5)
        rb = al.batch(batch); rb.run(); rb.sync()
        best = 1e9
        for _ in range(10):
            t0 = time.perf_counter(); rb.run(); rb.sync(); best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter(); sc, st, cg = al.align_batch(batch, scope == "full"); t_call = time.perf_counter() - t0
        t0 = time.perf_counter(); sc, st, cg = al.align_batch(batch, scope == "full"); t_call = min(t_call, time.perf_counter() - t0)
        print(f"{scope:5s} n={n:8d} run+sync {best * 1e6:9.1f} us ({n / best:.4g} aln/s)   align_batch (host in -> host out) {t_call * 1e6:9.1f} us ({n / t_call:.4g} aln/s)", flush=True)
        rb.close()
    al.close()
