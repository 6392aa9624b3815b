#!/usr/bin/env python3
"""Per-stage kernel times of the register-kernel cascade on C2-shaped data (development aid, no torch):
   python tools/stage_probe.py 245 24 3245      (N=pairs, default 10M; prints the library's stage lines)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pywfa_amd import datagen, _native
import common
n = int(os.environ.get("N", "10000000"))
L = int(os.environ.get("L", "150")); E = float(os.environ.get("E", "0.02"))
batch = datagen.generate(n, L, E, 1002)
_, nc = common.configs_pair(span="end-to-end", scope="score")
ref = None
for st in sys.argv[1:]:
    os.environ["WFA_HIP_FAST_STAGES"] = st
    al = _native.Aligner(nc); rb = al.batch(batch)
    os.environ["WFA_HIP_STAGE_TIMING"] = "0"
    rb.run(); rb.sync()
    for _ in range(5): rb.run()
    rb.sync()
    ms, pairs = rb.last_kernel()
    score, status, _ = rb.results(False)
    if ref is None: ref = (score.copy(), status.copy())
    same = bool((score == ref[0]).all() and (status == ref[1]).all())
    print(f"stages {st}: {ms:.3f} ms -> {n / ms / 1e3:.1f} M aln/s  same_as_first={same} checksum={int(score.sum())}", flush=True)
    os.environ["WFA_HIP_STAGE_TIMING"] = "1"
    sys.stderr.flush(); rb.run(); rb.sync()
    rb.close(); al.close()
