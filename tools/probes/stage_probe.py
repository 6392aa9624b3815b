#!/usr/bin/env python3
"""Stage timing of one exact configuration (development aid): python stage_probe.py L e n scope"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pywfa_amd import datagen, _native
import common
L, e, n, scope = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
kw = dict(span="end-to-end", scope=scope)
for a in sys.argv[5:]:
    k, v = a.split("="); kw[k] = int(v) if v.lstrip("-").isdigit() else v
batch = datagen.generate(n, L, e, 7)
oc, nc = common.configs_pair(**kw)
al = _native.Aligner(nc); rb = al.batch(batch)
rb.run(); rb.sync()
print(f"---- L={L} e={e} n={n} {kw}", file=sys.stderr, flush=True)
os.environ["X"] = "1"
t0 = time.time(); rb.run(); rb.sync(); wall = time.time() - t0
print(f"L={L} e={e} n={n} {scope}: {n / wall:.4g} aln/s wall {wall * 1e3:.2f} ms", flush=True)
rb.close(); al.close()
