#!/usr/bin/env python3
"""Parity of the exact mid-length path with the band pilot deciding (n >= 32768): a CPU-checked prefix (development aid)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import loader
from pywfa_amd import datagen
import common
bad_total = 0
for (L, e, n) in ((600, 0.10, 40000), (1000, 0.05, 40000), (1000, 0.01, 40000), (400, 0.12, 50000)):
    batch = datagen.generate(n, L, e, 31)
    for scope in ("score", "full"):
        oc, nc = common.configs_pair(span="end-to-end", scope=scope)
        full = scope == "full"
        score, status, cigars = common.gpu_run(nc, batch, full, resident=True)
        sub = datagen.subset(batch, np.arange(0, n, 40))
        o = loader.run(loader.oracle(), oc, sub, want_cigar=full)
        idx = np.arange(0, n, 40)
        bad = int(((score[idx] != o["score"]) | (status[idx] != o["status"])).sum())
        if full: bad += sum(1 for j, i in enumerate(idx) if cigars[i] != o["cigars"][j])
        bad_total += bad
        print(f"L={L} e={e} n={n} {scope}: {len(idx)} checked, {bad} bad", flush=True)
print("TOTAL BAD", bad_total)
