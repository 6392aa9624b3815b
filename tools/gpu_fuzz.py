#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the oracle (development aid): random configurations drawn from
what the library accepts, random read lengths / divergences / length differences, batches large enough to reach the
register kernels.  python tools/gpu_fuzz.py [rounds] [seed] [seconds]   (seconds: no new round is started after that many)"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import loader
from pywfa_amd import datagen, _native
import common

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # (the -m gpu slice runs on a time budget: tests/test_fuzz_gpu.py)
t_start = time.time()
rounds_done = 0
PEN = [(4, 6, 2), (4, 6, 2), (4, 6, 2), (4, 4, 2), (4, 6, 1), (3, 4, 1), (6, 5, 3), (5, 0, 3), (1, 1, 1), (2, 3, 1), (7, 3, 2),
       (5, 6, 2), (3, 5, 1), (2, 8, 1), (7, 11, 3), (9, 2, 4)]   # (round 4: shapes without an instantiation: compiled at run time)
bad_total = 0
for it in range(rounds):
    if budget_s and time.time() - t_start > budget_s and rounds_done >= 4:   # (at least four rounds whatever the budget: tests/test_fuzz_gpu.py asserts it)
        break
    rounds_done += 1
    x, o, e = PEN[int(rng.integers(len(PEN)))]
    kw = dict(mismatch=x, gap_opening=o, gap_extension=e, scope=str(rng.choice(["score", "full"])),
              span=str(rng.choice(["end-to-end", "ends-free"])))
    r = rng.random()
    if r < 0.15: kw["heuristic"] = "adaptive"
    elif r < 0.3: kw.update(heuristic="X-drop", xdrop=int(rng.choice([20, 100, 400])), steps_between_cutoffs=int(rng.choice([1, 1, 3])))
    if kw["span"] == "ends-free" and rng.random() < 0.3:
        kw.update(pattern_begin_free=int(rng.integers(0, 9)), pattern_end_free=int(rng.integers(0, 9)),
                  text_begin_free=int(rng.integers(0, 9)), text_end_free=int(rng.integers(0, 9)))
    if rng.random() < 0.2: kw["distance"] = "affine2p"
    if rng.random() < 0.3: kw["memory_mode"] = str(rng.choice(["medium", "low"]))
    if rng.random() < 0.1: kw["max_steps"] = int(rng.choice([6, 20, 60, 400]))
    if rng.random() < 0.12:   # round 4: match < 0 (mapped to the rescaled gap-affine form where nothing else reads the original; free begins + CIGAR are refused)
        kw["match"] = int(rng.choice([-1, -2, -3]))
        if kw["scope"] == "full":
            for k_ in ("pattern_begin_free", "pattern_end_free", "text_begin_free", "text_end_free"): kw.pop(k_, None)
    if rng.random() < 0.1:    # round 4: one-component distances (score scope runs on the gap-affine kernels)
        kw["distance"] = str(rng.choice(["indel", "levenshtein", "linear"]))
        for k_ in ("heuristic", "xdrop", "match"): kw.pop(k_, None)
    if rng.random() < 0.12:   # BiWFA: without heuristic or free ends (a step limit is honoured, round 3)
        for k_ in ("heuristic", "xdrop", "pattern_begin_free", "pattern_end_free", "text_begin_free", "text_end_free"): kw.pop(k_, None)
        kw["memory_mode"] = "biwfa"
    # a batch of mixed lengths and divergences, with length differences (end-to-end gaps)
    pats, txts = [], []
    nparts = int(rng.integers(1, 4))
    for part in range(nparts):
        L = int(rng.choice([8, 30, 64, 100, 150, 150, 150, 250, 400, 512, 600]))
        err = float(rng.choice([0.0, 0.01, 0.02, 0.02, 0.05, 0.1, 0.25]))
        n = int(rng.choice([300, 2000, 9000, 9000, 70000])) if L <= 250 else int(rng.choice([100, 700]))
        if it % 7 == 3 and part == 0: L, n = int(rng.choice([1500, 3000, 6000])), int(rng.choice([60, 160, 300]))  # long reads: banded kernel, split launches, wide-wavefront kernel
        if it % 21 == 10 and part == 0:
            L, n = int(rng.choice([22000, 30000])), int(rng.choice([40, 140]))
            if budget_s: n = 6   # (on a time budget: the oracle's exact gap-affine-2p runs at 30 kb take a second each)  # reads over 20 kb: 256-diagonal first window, the wide kernel's cut-off, piggy-back arena of the general kernel
        b = datagen.generate(n, L, err, int(rng.integers(1, 1 << 30)))
        cut = rng.integers(0, 20, size=n) * (rng.random(n) < 0.3)
        for i in (range(n) if n <= 9000 else range(0)):
            p, t = datagen.pair_strings(b, i)
            c = int(cut[i])
            if c and len(p) > c + 1:
                p = p[:-c] if i % 2 else p[c:]
            pats.append(p); txts.append(t)
    if pats:
        perm = rng.permutation(len(pats))
        batch = datagen.from_strings([pats[i] for i in perm], [txts[i] for i in perm])
    else:
        batch = b  # one large batch as generated (exercises the pilot that picks the first segment width)
    if it % 9 == 5:
        # a batch of >= 256 k pairs (the host-packed pipelined upload): a ragged head, some with letters outside ACGT, in
        # front of a generated body
        L = int(rng.choice([40, 100, 150, 250])); nb = int(rng.choice([270000, 400000]))
        body = datagen.generate(nb, L, float(rng.choice([0.01, 0.03, 0.08])), int(rng.integers(1, 1 << 30)))
        hp, ht = [], []
        for i in range(3000):
            p, t = datagen.pair_strings(body, i)
            k_ = i % 8
            if k_ == 0: p = p[:int(rng.integers(0, L))]
            if k_ == 1 and len(t) > 5: t = t[:3] + "N" + t[4:]
            if k_ == 2: p = p.lower()
            if k_ == 3: t = t + p[: int(rng.integers(0, 40))]
            hp.append(p); ht.append(t)
        head = datagen.from_strings(hp, ht, upper=False)
        shift = len(head["seqs"])
        batch = dict(seqs=np.concatenate([head["seqs"], body["seqs"]]),
                     p_off=np.concatenate([head["p_off"], body["p_off"][3000:] + shift]), p_len=np.concatenate([head["p_len"], body["p_len"][3000:]]),
                     t_off=np.concatenate([head["t_off"], body["t_off"][3000:] + shift]), t_len=np.concatenate([head["t_len"], body["t_len"][3000:]]))
        kw.pop("max_steps", None)
        if kw.get("memory_mode") == "biwfa": kw["memory_mode"] = "high"
    kw = common.clamp_free(kw, batch)
    try:
        oc, nc = common.configs_pair(**kw)
    except Exception as ex:
        print("skip", kw, ex); continue
    full = oc.scope == 1
    t0 = time.time()
    o_ = loader.run(loader.oracle(), oc, batch, want_cigar=full)
    score, status, cig = common.gpu_run(nc, batch, full, bool(it % 2))
    bad = int(((score != o_["score"]) | (status != o_["status"])).sum())
    if full and not bad:
        bad = sum(1 for i in range(len(score)) if bytes(cig[i]) != o_["cigars"][i])
    bad_total += bad
    print(f"{'OK ' if not bad else 'BAD'} round {it} n={len(score)} {kw} ({time.time() - t0:.1f}s)", flush=True)
    if bad:
        i = int(np.flatnonzero((score != o_["score"]) | (status != o_["status"]))[0]) if ((score != o_["score"]) | (status != o_["status"])).any() else next(i for i in range(len(score)) if bytes(cig[i]) != o_["cigars"][i])
        print("  first bad", i, o_["score"][i], score[i], o_["status"][i], status[i], datagen.pair_strings(batch, i))
print("ROUNDS", rounds_done, "of", rounds)
print("TOTAL BAD", bad_total)
sys.exit(1 if bad_total else 0)
