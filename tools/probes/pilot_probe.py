#!/usr/bin/env python3
"""150 bp rate per divergence and pilot threshold (WFA_HIP_PILOT_PCT: share of the sample the 16-diagonal stage may hand on) — development aid."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from pywfa_amd import datagen, _native
import common

n = 2_000_000
os.environ["WFA_HIP_STAGE_TIMING"] = "1"
for scope in ("score", "full"):
    for e in (0.02, 0.025, 0.03, 0.035, 0.04):
        batch = datagen.generate(n, 150, e, 77)
        oc, nc = common.configs_pair(span="end-to-end", scope=scope)
        ref = None
        for pct in (0, 40, 2):   # (0: the default)
            if pct: os.environ["WFA_HIP_PILOT_PCT"] = str(pct)
            else: os.environ.pop("WFA_HIP_PILOT_PCT", None)
            al = _native.Aligner(nc); rb = al.batch(batch); rb.run(); rb.sync()
            t0 = time.time()
            for _ in range(3): rb.run()
            rb.sync(); wall = (time.time() - t0) / 3
            score, status, _ = rb.results(False)
            if ref is None: ref = score.copy()
            print(f"{scope} e={e:.3f} pct={pct:3d} aln/s={n / wall:.4g} same={bool((score == ref).all())} general={rb.fallback_pairs()}", flush=True)
            rb.close(); al.close()
