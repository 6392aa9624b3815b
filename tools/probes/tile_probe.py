"""Tiled wide-wavefront kernel: parity on the ragged 2.5 kb corpus of tests/test_wide_gpu.py for several geometries, then timings."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from oracle import loader
from pywfa_amd import datagen
import test_wide_gpu as tw

def check(idx, env):
    for k, v in env.items(): os.environ[k] = v
    try:
        batch = tw.ragged_batch(180, 2500, 0.10, 9100 + idx)
        kw = common.clamp_free(dict(tw.CASES[idx]), batch)
        oc, nc = common.configs_pair(**kw)
        full = oc.scope == 1
        o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
        score, status, cigars = common.gpu_run(nc, batch, full, resident=True)
        try:
            common.assert_same(o, score, status, cigars, batch, f"tile {kw} {env}")
            print("ok  ", idx, env, flush=True)
        except AssertionError as e:
            print("FAIL", idx, env, str(e)[:300], flush=True)
    finally:
        for k in env: del os.environ[k]

which = sys.argv[1:] or ["parity"]
if "parity" in which:
    for idx in range(len(tw.CASES)):
        check(idx, {})
    for idx in (1, 2, 11, 12):
        for env in ({"WFA_HIP_TILE_T": "4", "WFA_HIP_TILE_WT": "64"}, {"WFA_HIP_TILE_T": "16", "WFA_HIP_TILE_WT": "128"},
                    {"WFA_HIP_TILE_T": "8", "WFA_HIP_TILE_WT": "256", "WFA_HIP_TILE_THREADS": "128"}, {"WFA_HIP_TILE_T": "2", "WFA_HIP_TILE_WT": "64", "WFA_HIP_TILE_THREADS": "512"}):
            check(idx, env)
