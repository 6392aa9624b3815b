#!/usr/bin/env python3
"""Development aid: 150 bp score-only under a heuristic / free ends, first stage forced (WFA_HIP_LANE_HEUR = 0 / 1 / 2, default: the pilot)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import loader
from pywfa_amd import datagen, _native
import common

n = int(os.environ.get("N", "2000000"))
CFG = {"adapt": dict(span="end-to-end", scope="score", heuristic="adaptive"),
       "ef": dict(span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2, scope="score"),
       "ef2": dict(span="ends-free", pattern_begin_free=2, pattern_end_free=7, text_begin_free=1, text_end_free=2, scope="score"),
       "efend": dict(span="ends-free", pattern_begin_free=0, pattern_end_free=20, text_begin_free=0, text_end_free=20, scope="score"),
       "steps": dict(span="end-to-end", scope="score", max_steps=30),
       "xdrop": dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=100),
       "xdrop20": dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=20)}
for name in os.environ.get("WHICH", "adapt").split():
    for e in [float(x) for x in os.environ.get("E", "0.02").split()]:
        batch = datagen.generate(n, 150, e, 1002)
        kw = common.clamp_free(CFG[name], batch)
        oc, nc = common.configs_pair(**kw)
        sub = datagen.subset(batch, np.arange(20000))
        o = loader.run(loader.oracle(), oc, sub)
        for forced in os.environ.get("FORCED", "- 0 1 2").split():
            if forced == "-": os.environ.pop("WFA_HIP_LANE_HEUR", None)
            else: os.environ["WFA_HIP_LANE_HEUR"] = forced
            al = _native.Aligner(nc); rb = al.batch(batch); rb.run(); rb.sync()
            t0 = time.time()
            for _ in range(3): rb.run()
            rb.sync(); wall = (time.time() - t0) / 3
            ms, _ = rb.last_kernel()
            score, status, _ = rb.results(False)
            bad = int(((score[:20000] != o["score"]) | (status[:20000] != o["status"])).sum())
            print(f"{name} e={e} LANE_HEUR={forced}: kernel_ms={ms:.3f} aln/s={n / wall:.4g} bad={bad} general={rb.fallback_pairs()}", flush=True)
            rb.close(); al.close()
