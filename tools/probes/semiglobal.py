#!/usr/bin/env python3
"""Semi-global use (a short pattern somewhere inside a longer text: ends-free on the text): rate by text length (development aid)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pywfa_amd import datagen, _native
import common
rng = np.random.default_rng(1)
for (pl, flank, n) in ((150, 50, 200000), (150, 400, 100000), (150, 2000, 20000), (1000, 2000, 20000)):
    b = datagen.generate(n, pl, 0.03, 5)
    pats, txts = [], []
    fl = "".join(rng.choice(list("ACGT"), size=2 * flank * 64))
    for i in range(n):
        p, t = datagen.pair_strings(b, i)
        o = (i * 37) % (len(fl) - 2 * flank)
        pats.append(p); txts.append(fl[o:o + flank] + t + fl[o + flank:o + 2 * flank])
    batch = datagen.from_strings(pats, txts)
    for scope in ("score", "full"):
        for heur in (None, "adaptive"):
            kw = dict(span="ends-free", text_begin_free=flank + 20, text_end_free=flank + 20, scope=scope)
            if heur: kw["heuristic"] = heur
            oc, nc = common.configs_pair(**kw)
            al = _native.Aligner(nc); rb = al.batch(batch)
            rb.run(); rb.sync()
            t0 = time.time(); rb.run(); rb.sync(); wall = time.time() - t0
            fb = rb.fallback_pairs()
            rb.close(); al.close()
            print(f"pattern {pl} in text {pl + 2 * flank}, {scope:5s} heuristic={heur}: n={n} {n / wall:.4g} aln/s  general={fb}", flush=True)
