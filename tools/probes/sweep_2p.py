#!/usr/bin/env python3
"""gap-affine-2p over read length x divergence, exact and wf-adaptive, score and full (development aid; GPU box)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pywfa_amd import datagen, _native
import common
bases = int(os.environ.get("SWEEP_BASES", "200000000"))
for heur in (None, "adaptive"):
    for scope in ("score", "full"):
        for L in (150, 300, 600, 1000, 2000, 5000):
            for e in (0.02, 0.08):
                n = max(64, min(1000000, bases // (2 * L)))
                if heur is None and L >= 2000: n = max(64, n // 8)
                batch = datagen.generate(n, L, e, 9)
                kw = dict(distance="affine2p", span="end-to-end", scope=scope)
                if heur: kw["heuristic"] = heur
                oc, nc = common.configs_pair(**kw)
                al = _native.Aligner(nc); rb = al.batch(batch)
                rb.run(); rb.sync()
                t0 = time.time(); rb.run(); rb.sync(); wall = time.time() - t0
                fb = rb.fallback_pairs()
                rb.close(); al.close()
                print(f"2p heur={heur} {scope:5s} L={L:5d} e={e:.2f} n={n:7d}  {n / wall:12.4g} aln/s  {2 * L * n / wall / 1e9:7.2f} Gbases/s  general={fb}", flush=True)
