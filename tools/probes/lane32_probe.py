#!/usr/bin/env python3
"""150 bp score-only rate per divergence and first stage of the cascade (development aid): the pilot's pick against a forced order."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from pywfa_amd import datagen, _native
import common

n = 2_000_000
for e in (0.02, 0.03, 0.04, 0.05, 0.06, 0.08, 0.10):
    batch = datagen.generate(n, 150, e, 77)
    oc, nc = common.configs_pair(span="end-to-end", scope="score")
    ref = None
    for st in ("", "189", "a9", "a89", "89", "1a9", "9"):
        if st: os.environ["WFA_HIP_FAST_STAGES"] = st
        else: os.environ.pop("WFA_HIP_FAST_STAGES", None)
        al = _native.Aligner(nc); rb = al.batch(batch); rb.run(); rb.sync()
        t0 = time.time()
        for _ in range(3): rb.run()
        rb.sync(); wall = (time.time() - t0) / 3
        score, status, _ = rb.results(False)
        if ref is None: ref = score.copy()
        print(f"e={e:.2f} stages={st or 'pilot':6s} aln/s={n / wall:.4g} same={bool((score == ref).all())} general={rb.fallback_pairs()}", flush=True)
        rb.close(); al.close()
