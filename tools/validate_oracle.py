#!/usr/bin/env python3
"""Pin oracle/wfa_oracle.c against the real reference (oracle/_ref = WFA2-lib v2.3 compiled from
/root/reference): bit-exact (status, score, op string) on large random corpora per configuration.

Runs only where oracle/_ref exists.  Usage: python tools/validate_oracle.py [--pairs N] [--quick]
Prints one line per (corpus, config) and exits non-zero on any mismatch.
"""
import argparse
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import loader  # noqa: E402
from pywfa_amd import datagen  # noqa: E402


def corpus_special(seed=7):
    """Edge cases: empty / length-1 / identical / unrelated / N / lower-case / tie-heavy repeats."""
    rng = np.random.default_rng(seed)
    pats, txts = [], []

    def rnd(n, alphabet="ACGT"):
        return "".join(rng.choice(list(alphabet), size=n))

    def mutate(s, e):
        out = []
        for ch in s:
            u = rng.random()
            if u < e / 3:
                out.append(rng.choice([c for c in "ACGT" if c != ch]))
            elif u < 2 * e / 3:
                out.append(rng.choice(list("ACGT"))); out.append(ch)
            elif u < e:
                pass
            else:
                out.append(ch)
        return "".join(out)

    pats += ["", "ACGT", "", "A", "A", "AC"]; txts += ["ACGT", "", "", "A", "C", "CA"]
    for n in (1, 2, 3, 5, 17, 64, 150):
        s = rnd(n)
        pats.append(s); txts.append(s)              # identical
        pats.append(rnd(n)); txts.append(rnd(n))    # unrelated
        pats.append(s); txts.append(s[::-1])
    for _ in range(300):                            # homopolymers / tandem repeats (ties)
        unit = rnd(int(rng.integers(1, 5)))
        a = unit * int(rng.integers(3, 40))
        b = mutate(a, 0.15) + unit * int(rng.integers(0, 5))
        pats.append(a[:200]); txts.append(b[:220])
    for _ in range(200):                            # low complexity 2-letter alphabet
        a = rnd(int(rng.integers(10, 200)), "AT")
        pats.append(a); txts.append(mutate(a, 0.2))
    for _ in range(100):                            # N-containing / non-ACGT letters
        a = rnd(int(rng.integers(20, 120)), "ACGTN")
        pats.append(a); txts.append(mutate(a, 0.1).replace("G", "R", 1))
    for _ in range(200):                            # long gaps (second affine piece)
        a = rnd(int(rng.integers(80, 400)))
        cut = int(rng.integers(10, 60)); pos = int(rng.integers(0, len(a) - cut))
        if rng.random() < 0.5:
            b = a[:pos] + a[pos + cut:]
        else:
            b = a[:pos] + rnd(cut) + a[pos:]
        pats.append(a); txts.append(mutate(b, 0.04))
    for _ in range(200):                            # text is a window of a longer reference
        ref = rnd(int(rng.integers(100, 600)))
        lo = int(rng.integers(0, len(ref) // 2)); hi = int(rng.integers(lo + 5, len(ref)))
        pats.append(mutate(ref[lo:hi], 0.05)); txts.append(ref)
    return datagen.from_strings(pats, txts)


def configs(quick):
    base = []
    for distance in ("affine", "affine2p"):
        for span, free in (("end-to-end", (0, 0, 0, 0)), ("ends-free", (0, 0, 0, 0)),
                           ("ends-free", (8, 7, 3, 2)), ("ends-free", (0, 5, 0, 9)),
                           ("ends-free", (20, 0, 20, 0))):
            for heur in (None, "adaptive", ("X-drop", 20), ("X-drop", 100), ("X-drop", 1000)):
                for scope in ("score", "full"):
                    for max_steps in (0, 10):
                        kw = dict(distance=distance, span=span, scope=scope, max_steps=max_steps,
                                  pattern_begin_free=free[0], pattern_end_free=free[1],
                                  text_begin_free=free[2], text_end_free=free[3])
                        if isinstance(heur, tuple):
                            kw.update(heuristic=heur[0], xdrop=heur[1])
                        else:
                            kw.update(heuristic=heur)
                        base.append(kw)
    extra = [
        dict(distance="affine", mismatch=2, gap_opening=3, gap_extension=1),
        dict(distance="affine", mismatch=5, gap_opening=0, gap_extension=3),
        dict(distance="affine", mismatch=1, gap_opening=1, gap_extension=1, span="end-to-end"),
        dict(distance="affine2p", mismatch=5),
        dict(distance="affine2p", mismatch=3, gap_opening=4, gap_extension=2, gap_opening2=12, gap_extension2=1),
        dict(distance="affine", match=-1, span="end-to-end"),
        dict(distance="affine", match=-2, mismatch=3, span="end-to-end", scope="score"),
        dict(distance="affine2p", match=-1, span="ends-free", pattern_end_free=5, text_end_free=5),
        dict(distance="affine", match=-1, heuristic="X-drop", xdrop=100, span="end-to-end"),
        dict(distance="affine", heuristic="adaptive", min_wavefront_length=5, max_distance_threshold=10, steps_between_cutoffs=3),
        dict(distance="affine", heuristic="X-drop", xdrop=50, steps_between_cutoffs=4),
        dict(distance="affine", memory_mode="medium"),
        dict(distance="affine2p", memory_mode="low", span="end-to-end"),
        dict(distance="affine", wildcard="N"),
    ]
    if quick:
        base = base[::7]
    return base + extra


def clamp_free(kw, batch):
    """Ends-free sizes must not exceed the sequence lengths (the reference exit(1)s)."""
    pl, tl = int(batch["p_len"].min()), int(batch["t_len"].min())
    kw = dict(kw)
    for k, lim in (("pattern_begin_free", pl), ("pattern_end_free", pl),
                   ("text_begin_free", tl), ("text_end_free", tl)):
        if kw.get(k, 0) > lim:
            kw[k] = lim
    return kw


def _ref_child(kw, batch):
    return loader.run(loader.reference(), loader.make_config(**kw), batch)


def run_reference(kw, batch):
    """The reference exit(1)s on some heuristic runs ("Maximum allocated wavefronts reached",
    wavefront_compute.c:425-428): run it in a child process so that the sweep survives."""
    from concurrent.futures import ProcessPoolExecutor
    from concurrent.futures.process import BrokenProcessPool
    import multiprocessing as mp
    try:
        with ProcessPoolExecutor(max_workers=1, mp_context=mp.get_context("fork")) as ex:
            return ex.submit(_ref_child, kw, batch).result()
    except BrokenProcessPool:
        return None


def compare(name, kw, batch):
    kw = clamp_free(kw, batch)
    cfg = loader.make_config(**kw)
    t0 = time.time()
    r = run_reference(kw, batch)
    if r is None:
        print(f"SKIP {name:14s} reference process exited (exit(1) inside WFA2-lib) {kw}", flush=True)
        return 0
    t1 = time.time()
    o = loader.run(loader.oracle(), cfg, batch)
    t2 = time.time()
    bad = np.flatnonzero((r["score"] != o["score"]) | (r["status"] != o["status"]))
    nb = bad.size
    first = int(bad[0]) if nb else -1
    if r["cigars"] is not None:
        cb = [i for i, (a, b) in enumerate(zip(r["cigars"], o["cigars"])) if a != b]
        nb += len(cb)
        if first < 0 and cb:
            first = cb[0]
    n = len(batch["p_len"])
    dropped = int((r["status"] != 0).sum())
    print(f"{'OK ' if nb == 0 else 'BAD'} {name:14s} n={n:6d} mism={nb:5d} nonzero_status={dropped:6d} "
          f"ref={t1 - t0:6.2f}s oracle={t2 - t1:6.2f}s {kw}", flush=True)
    if nb:
        p, t = datagen.pair_strings(batch, first)
        print("   first bad pair", first, "ref", r["score"][first], r["status"][first],
              (r["cigars"] or [None] * n)[first], "oracle", o["score"][first], o["status"][first],
              (o["cigars"] or [None] * n)[first])
        print("   P", p); print("   T", t)
    return nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2000)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--big", action="store_true", help="also run 100k-pair corpora on the main configs")
    args = ap.parse_args()
    loader.build()
    corpora = [("special", corpus_special())]
    for L, e in ((150, 0.005), (150, 0.02), (150, 0.05), (150, 0.15), (1000, 0.02), (1000, 0.08), (1000, 0.2)):
        n = args.pairs if L <= 150 else max(50, args.pairs // 10)
        corpora.append((f"L{L}_e{e}", datagen.generate(n, L, e, 4242 + L)))
    corpora.append(("L10000_e0.08", datagen.generate(20 if args.quick else 60, 10000, 0.08, 1003)))
    total_bad = 0
    cfgs = configs(args.quick)
    for name, batch in corpora:
        for kw in cfgs:
            if name.startswith("L10000") and kw.get("heuristic") is None and kw.get("distance") == "affine2p" and kw.get("scope", "full") == "full" and kw.get("max_steps", 0) == 0:
                continue  # 400 MB of history per pair: covered at 1 kb
            total_bad += compare(name, kw, batch)
    if args.big:
        big = datagen.generate(100000, 150, 0.02, 1002)
        for kw in (dict(span="end-to-end", scope="score"), dict(span="end-to-end", scope="full"),
                   dict(scope="full"), dict(distance="affine2p", scope="full"),
                   dict(heuristic="adaptive"), dict(heuristic="X-drop", xdrop=100)):
            total_bad += compare("C2x100k", kw, big)
    print("TOTAL MISMATCHES", total_bad)
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
