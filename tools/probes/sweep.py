#!/usr/bin/env python3
"""Exact (pywfa's default: no heuristic) gap-affine alignment over read length x divergence: looking for cliffs between the kernel
families (development aid; run on the GPU box).  Device time only; parity is the test suite's job."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pywfa_amd import datagen, _native
import common

bases = int(os.environ.get("SWEEP_BASES", "300000000"))   # bases per batch (both sequences)
for scope in ("score", "full"):
    for L in (150, 300, 600, 1000, 2000, 4000):
        for e in (0.01, 0.05, 0.10):
            n = max(256, min(2000000, bases // (2 * L)))
            batch = datagen.generate(n, L, e, 7)
            oc, nc = common.configs_pair(span="end-to-end", scope=scope)
            al = _native.Aligner(nc); rb = al.batch(batch)
            rb.run(); rb.sync()
            t0 = time.time(); rb.run(); rb.sync(); wall = time.time() - t0
            ms, _ = rb.last_kernel()
            fb = rb.fallback_pairs()
            rb.close(); al.close()
            print(f"{scope:5s} L={L:5d} e={e:.2f} n={n:8d}  {n / wall:12.4g} aln/s  {2 * L * n / wall / 1e9:8.2f} Gbases/s  kernel {ms:8.2f} ms  general={fb}", flush=True)
