#!/usr/bin/env python3
"""Compare the piggy-back and the explicit history of the banded kernel on a few long pairs (development aid)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import loader
from pywfa_amd import datagen
import common
L = int(os.environ.get("L", "3000")); n = int(os.environ.get("N", "64")); e = float(os.environ.get("E", "0.05"))
kw = dict(span="end-to-end", scope="full")
if os.environ.get("ADAPT"): kw["heuristic"] = "adaptive"
batch = datagen.generate(n, L, e, 55)
oc, nc = common.configs_pair(**kw)
o = loader.run(loader.oracle(), oc, batch)
for pb in ("0", "1"):
    os.environ["WFA_HIP_BAND_PB"] = pb
    score, status, cig = common.gpu_run(nc, batch, True, True)
    bad = [i for i in range(n) if cig[i] != o["cigars"][i] or score[i] != o["score"][i]]
    print("pb", pb, "bad", len(bad), "of", n)
    for i in bad[:2]:
        a, b = common.rle(o["cigars"][i]), common.rle(cig[i])
        j = next((x for x in range(min(len(a), len(b))) if a[x] != b[x]), min(len(a), len(b)))
        print("  pair", i, "score", o["score"][i], score[i], "len", len(o["cigars"][i]), len(cig[i]))
        print("   ref ", a[max(0, j - 40): j + 60])
        print("   gpu ", b[max(0, j - 40): j + 60])
