#!/usr/bin/env python3
"""wf-adaptive gap-affine alignment (full CIGAR) over read length x divergence: looking for cliffs (development aid; GPU box)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pywfa_amd import datagen, _native
import common
bases = int(os.environ.get("SWEEP_BASES", "400000000"))
for scope in ("full", "score"):
    for L in (150, 300, 600, 1000, 2000, 5000, 20000, 50000):
        for e in (0.02, 0.08, 0.15):
            n = max(64, min(2000000, bases // (2 * L)))
            batch = datagen.generate(n, L, e, 9)
            oc, nc = common.configs_pair(span="end-to-end", scope=scope, heuristic="adaptive")
            al = _native.Aligner(nc); rb = al.batch(batch)
            rb.run(); rb.sync()
            t0 = time.time(); rb.run(); rb.sync(); wall = time.time() - t0
            ms, _ = rb.last_kernel(); fb = rb.fallback_pairs()
            rb.close(); al.close()
            print(f"{scope:5s} L={L:6d} e={e:.2f} n={n:8d}  {n / wall:12.4g} aln/s  {2 * L * n / wall / 1e9:8.2f} Gbases/s  kernel {ms:8.2f} ms  general={fb}", flush=True)
