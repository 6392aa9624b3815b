#!/usr/bin/env python3
"""Development aid: C2 kernel time under different stage orders (WFA_HIP_FAST_STAGES) + a parity check of a sample."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import loader
from pywfa_amd import datagen, _native
import common

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
variants = sys.argv[2:] or ["689", "189"]
batch = datagen.generate(n, 150, 0.02, 1002)
oc, nc = common.configs_pair(span="end-to-end", scope="score")
ns = min(n, 200000)
o = loader.run(loader.reference() if loader.have_reference() else loader.oracle(), oc, datagen.subset(batch, np.arange(ns)), want_cigar=False)
for v in variants:
    env = dict(kv.split("=") for kv in v.split(",")[1:]) if "," in v else {}
    stages = v.split(",")[0]
    os.environ["WFA_HIP_FAST_STAGES"] = stages
    for k_, v_ in env.items(): os.environ[k_] = v_
    al = _native.Aligner(nc); rb = al.batch(batch)
    rb.run(); rb.sync()
    t0 = time.time()
    for _ in range(5): rb.run()
    rb.sync(); wall = (time.time() - t0) / 5
    ms, _ = rb.last_kernel()
    score, status, _ = rb.results(False)
    bad = int(((score[:ns] != o["score"]) | (status[:ns] != o["status"])).sum())
    print(f"stages {stages:6s} {env} kernel_ms={ms:8.3f} wall_ms={wall*1e3:8.3f} G aln/s={n/wall/1e9:6.3f} mismatches(first {ns})={bad} fallback={rb.fallback_pairs()} nonzero_status={int((status!=0).sum())}", flush=True)
    rb.close(); al.close()
    for k_ in env: os.environ.pop(k_, None)
