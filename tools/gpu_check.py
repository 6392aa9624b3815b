#!/usr/bin/env python3
"""Quick GPU-vs-oracle sweep (development aid; the real parity tests live in tests/).

python tools/gpu_check.py [--pairs N] [--quick]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import loader  # noqa: E402
from pywfa_amd import datagen, _native  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
import validate_oracle as vo  # noqa: E402


def native_config(kw):
    oc = loader.make_config(**kw)
    c = _native.Config()
    for name, _ in _native.Config._fields_:
        setattr(c, name, getattr(oc, name))
    return oc, c


def compare(name, kw, batch, resident=False):
    kw = vo.clamp_free(kw, batch)
    oc, nc = native_config(kw)
    full = oc.scope == 1
    t0 = time.time()
    o = loader.run(loader.oracle(), oc, batch)
    t1 = time.time()
    al = _native.Aligner(nc)
    if resident:
        rb = al.batch(batch)
        rb.run(); rb.sync()
        score, status, cig = rb.results(full)
        ms, pairs = rb.last_kernel()
        rb.close()
    else:
        score, status, cig = al.align_batch(batch, full)
        ms = -1
    t2 = time.time()
    al.close()
    bad = np.flatnonzero((score != o["score"]) | (status != o["status"]))
    nb = bad.size
    first = int(bad[0]) if nb else -1
    n = len(batch["p_len"])
    if full:
        ops, cbeg, clen = cig
        for i in range(n):
            g = ops[cbeg[i]:cbeg[i] + clen[i]].tobytes()
            if g != o["cigars"][i]:
                nb += 1
                if first < 0:
                    first = i
    print(f"{'OK ' if nb == 0 else 'BAD'} {name:14s} n={n:6d} mism={nb:5d} oracle={t1 - t0:6.2f}s gpu={t2 - t1:6.2f}s "
          f"kernel_ms={ms:.3f} {kw}", flush=True)
    if nb:
        p, t = datagen.pair_strings(batch, first)
        gc = ops[cbeg[first]:cbeg[first] + clen[first]].tobytes() if full else None
        print("   first bad pair", first, "oracle", o["score"][first], o["status"][first],
              (o["cigars"] or [None] * n)[first], "gpu", score[first], status[first], gc)
        print("   P", p[:300]); print("   T", t[:300])
    return nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=500)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--stride", type=int, default=7)
    args = ap.parse_args()
    print("devices:", _native.lib().wfa_hip_device_count())
    corpora = [("special", vo.corpus_special())]
    for L, e in ((150, 0.02), (150, 0.15), (1000, 0.08)):
        n = args.pairs if L <= 150 else max(50, args.pairs // 10)
        corpora.append((f"L{L}_e{e}", datagen.generate(n, L, e, 4242 + L)))
    corpora.append(("L10000_e0.08", datagen.generate(8, 10000, 0.08, 1003)))
    cfgs = vo.configs(True)
    if args.quick:
        cfgs = cfgs[::args.stride]
    total = 0
    for name, batch in corpora:
        for i, kw in enumerate(cfgs):
            if name.startswith("L10000") and kw.get("heuristic") is None and kw.get("distance") == "affine2p" and kw.get("scope", "full") == "full" and kw.get("max_steps", 0) == 0:
                continue
            total += compare(name, kw, batch, resident=(i % 2 == 0))
    print("TOTAL MISMATCHES", total)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
