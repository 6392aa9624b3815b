#!/usr/bin/env python3
"""Two-round (lazy) against one-round extension of the 32-lane segments on 150 bp score-only batches (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from pywfa_amd import datagen, _native
import common
n = 2_000_000
for e in (0.01, 0.02, 0.04):
    batch = datagen.generate(n, 150, e, 77)
    oc, nc = common.configs_pair(span="end-to-end", scope="score")
    for st in ("89", "45", "8", "4"):
        os.environ["WFA_HIP_FAST_STAGES"] = st
        al = _native.Aligner(nc); rb = al.batch(batch); rb.run(); rb.sync()
        t0 = time.time()
        for _ in range(3): rb.run()
        rb.sync(); wall = (time.time() - t0) / 3
        print(f"e={e:.2f} stages={st:3s} aln/s={n / wall:.4g} general={rb.fallback_pairs()}", flush=True)
        rb.close(); al.close()
