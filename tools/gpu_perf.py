#!/usr/bin/env python3
"""Throughput of the BASELINE configurations beyond C2 (development aid + numbers for DESIGN.md)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import loader
from pywfa_amd import datagen, _native
import common

def run(label, n, L, e, seed, kw, cpu_n=None, reps=2, check=True, trim=0):
    batch = datagen.generate(n, L, e, seed)
    if trim: batch = datagen.trim_text(batch, trim)   # (bench.py's C4 configurations: 50 bases cut off both ends of every text)
    kw = common.clamp_free(kw, batch)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    al = _native.Aligner(nc); t0 = time.time(); rb = al.batch(batch); t_up = time.time() - t0
    rb.run(); rb.sync()
    t0 = time.time()
    for _ in range(reps): rb.run()
    rb.sync(); wall = (time.time() - t0) / reps
    ms, pairs = rb.last_kernel()
    score, status, cig = rb.results(full)
    fb = rb.fallback_pairs(); alg = rb.algorithmic_bytes(); rb.close(); al.close()
    cpu_n = min(n, cpu_n or n)
    if os.environ.get("NO_CPU"):   # profile passes: the device work only
        print(f"{label:42s} kernel_ms={ms:9.3f} aln/s={n / wall:.4g} handed_to_general={fb} pairs={n} runs={reps + 1}", flush=True)
        return None
    sub = datagen.subset(batch, np.arange(cpu_n))
    fn = loader.reference() if loader.have_reference() else loader.oracle()
    t0 = time.time(); o = loader.run(fn, oc, sub); t_cpu = time.time() - t0
    bad = int(((score[:cpu_n] != o["score"]) | (status[:cpu_n] != o["status"])).sum())
    if full and check:
        ops, cbeg, clen = cig
        bad += sum(1 for i in range(cpu_n) if ops[cbeg[i]:cbeg[i]+clen[i]].tobytes() != o["cigars"][i])
    rec = dict(label=label, n=n, L=L, e=e, kernel_ms=ms, wall_ms=wall*1e3, aln_per_s=n/wall, cpu_aln_per_s=cpu_n/t_cpu,
               speedup_vs_1thread=(n/wall)/(cpu_n/t_cpu), mismatches=bad, fallback=fb, mean_score=float(score.mean()),
               nonzero_status=int((status != 0).sum()), upload_s=t_up, cfg=kw)
    if os.environ.get("BRIEF"):
        print(f"{label:42s} kernel_ms={ms:9.3f} aln/s={n / wall:.4g} mism={bad} handed_to_general={fb}", flush=True)
    else:
        print(json.dumps(rec), flush=True)
    return rec

which = sys.argv[1:] or ["C1", "C3", "C3s", "C4a", "C5", "E5", "E10"]
if "C1" in which: run("C1 150bp full", 1000000, 150, 0.02, 1001, dict(scope="full"), cpu_n=200000)
if "E5" in which: run("150bp 5% score", 1000000, 150, 0.05, 1002, dict(span="end-to-end", scope="score"), cpu_n=100000)
if "E10" in which: run("150bp 10% full", 500000, 150, 0.10, 1002, dict(scope="full"), cpu_n=50000)
if "C3" in which: run("C3 10kb adaptive full", 100000, 10000, 0.08, 1003, dict(span="end-to-end", scope="full", heuristic="adaptive"), cpu_n=400)
if "C3s" in which: run("10kb adaptive score", 100000, 10000, 0.08, 1003, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=400)
if "C1big" in which: run("C1 150bp full, 10M pairs", 10000000, 150, 0.02, 1001, dict(scope="full"), cpu_n=200000, reps=2)
if "C3big" in which: run("C3 10kb adaptive full, 400k pairs", 400000, 10000, 0.08, 1003, dict(span="end-to-end", scope="full", heuristic="adaptive"), cpu_n=100, reps=2)
if "C4abig" in which: run("C4 adaptive full, 100k pairs", 100000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive"), cpu_n=50, reps=2)
if "C4ax5" in which: run("C4 adaptive full, mismatch=5 (run-time shape), 100k pairs", 100000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive", mismatch=5), cpu_n=50, reps=2)
if "C4bench" in which: run("C4 adaptive full as bench.py runs it (texts trimmed by 50), 100k pairs", 100000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive"), cpu_n=50, reps=2, trim=50)
if "C4sbig" in which: run("C4 adaptive score, 100k pairs", 100000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="score", heuristic="adaptive"), cpu_n=50, reps=2)
if "C3x8k" in which: run("10kb exact score, 8192 pairs", 8192, 10000, 0.08, 1003, dict(span="end-to-end", scope="score"), cpu_n=16, reps=1)
if "C3xf8k" in which: run("10kb exact full, 8192 pairs", 8192, 10000, 0.08, 1003, dict(span="end-to-end", scope="full"), cpu_n=16, reps=1)
if "C4x4k" in which: run("C4 exact full, 4096 pairs", 4096, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full"), cpu_n=4, reps=1)
if "C3x20" in which: run("10kb 20% exact score", 1024, 10000, 0.20, 1003, dict(span="end-to-end", scope="score"), cpu_n=8, reps=1)
if "C3x" in which: run("10kb exact score", 2000, 10000, 0.08, 1003, dict(span="end-to-end", scope="score"), cpu_n=40)
if "C4a" in which: run("C4 10kb affine2p endsfree adaptive full", 10000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive"), cpu_n=200)
if "C4m" in which: run("C4 10kb affine2p endsfree adaptive full, memory_mode=medium", 10000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive", memory_mode="medium"), cpu_n=200)
if "C4s" in which: run("C4 10kb affine2p endsfree adaptive score", 10000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="score", heuristic="adaptive"), cpu_n=200)
if "C4xs" in which: run("C4 10kb affine2p endsfree EXACT score", 1024, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="score"), cpu_n=8, reps=1)
if "C4x8" in which: run("C4 exact full, 8 pairs", 8, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full"), cpu_n=2, reps=1)
if "C3x8" in which: run("10kb exact full, 8 pairs", 8, 10000, 0.08, 1003, dict(span="end-to-end", scope="full"), cpu_n=4, reps=1)
if "C3a8" in which: run("10kb adaptive full, 8 pairs", 8, 10000, 0.08, 1003, dict(span="end-to-end", scope="full", heuristic="adaptive"), cpu_n=8, reps=1)
if "C4x" in which: run("C4 10kb affine2p endsfree EXACT full", 1024, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full"), cpu_n=8, reps=1)
if "C3xf" in which: run("10kb affine EXACT full", 1024, 10000, 0.08, 1003, dict(span="end-to-end", scope="full"), cpu_n=16, reps=1)
if "C5" in which: run("C5 100kb xdrop", 2000, 100000, 0.08, 1005, dict(span="end-to-end", scope="full", heuristic="X-drop", xdrop=20), cpu_n=100)
if "X150" in which: run("150bp X-drop(100) score", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=100), cpu_n=100000)
if "X150l" in which: run("150bp X-drop(100) score, 0.5 %", 2000000, 150, 0.005, 1002, dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=100), cpu_n=100000)
if "X20" in which: run("150bp X-drop(20) score", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=20), cpu_n=100000)
if "X20l" in which: run("150bp X-drop(20) score, 0.5 %", 2000000, 150, 0.005, 1002, dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=20), cpu_n=100000)
if "A150m" in which: run("150bp adaptive score, 1 %", 2000000, 150, 0.01, 1002, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=100000)
if "S150" in which: run("150bp step limit 30, score", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score", max_steps=30), cpu_n=100000)
if "A150l" in which: run("150bp adaptive score, 0.5 %", 2000000, 150, 0.005, 1002, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=100000)
if "A150" in which: run("150bp adaptive score", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=100000)
if "N150" in which: run("150bp no heuristic score", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score"), cpu_n=100000)
if "B10k" in which: run("10kb BiWFA full", 2000, 10000, 0.08, 1003, dict(span="end-to-end", scope="full", memory_mode="biwfa"), cpu_n=40)
if "B10kbig" in which: run("10kb BiWFA full, 16384 pairs", 16384, 10000, 0.08, 1003, dict(span="end-to-end", scope="full", memory_mode="biwfa"), cpu_n=16, reps=1)
if "B100k" in which: run("100kb BiWFA full, 256 pairs", 256, 100000, 0.08, 1005, dict(span="end-to-end", scope="full", memory_mode="biwfa"), cpu_n=1, reps=1)
if "B1k" in which: run("1kb BiWFA full", 50000, 1000, 0.08, 1003, dict(span="end-to-end", scope="full", memory_mode="biwfa"), cpu_n=2000)
if "A30k" in which: run("30kb adaptive full", 4096, 30000, 0.08, 1005, dict(span="end-to-end", scope="full", heuristic="adaptive"), cpu_n=8, reps=1)
if "C5a8k" in which: run("100kb adaptive full, 8192 pairs", 8192, 100000, 0.08, 1005, dict(span="end-to-end", scope="full", heuristic="adaptive"), cpu_n=4, reps=1)
if "C5as8k" in which: run("100kb adaptive score, 8192 pairs", 8192, 100000, 0.08, 1005, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=4, reps=1)
if "X100k" in which: run("100kb exact score", 64, 100000, 0.08, 1005, dict(span="end-to-end", scope="score"), cpu_n=1, reps=1)
if "X100kf" in which: run("100kb exact full", 64, 100000, 0.08, 1005, dict(span="end-to-end", scope="full"), cpu_n=1, reps=1)
if "X30k" in which: run("30kb exact score", 512, 30000, 0.08, 1005, dict(span="end-to-end", scope="score"), cpu_n=2, reps=1)
if "X30kf" in which: run("30kb exact full", 512, 30000, 0.08, 1005, dict(span="end-to-end", scope="full"), cpu_n=2, reps=1)
if "EF150" in which: run("150bp ends-free(8,7,3,2) score", 2000000, 150, 0.02, 1002, dict(span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2, scope="score"), cpu_n=100000)
# round 4: penalties without an instantiation (run-time compiled kernels), configurations mapped to gap-affine + score translation
if "M5" in which: run("150bp mismatch=5 score", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score", mismatch=5), cpu_n=100000)
if "M5f" in which: run("150bp mismatch=5 full", 1000000, 150, 0.02, 1002, dict(scope="full", mismatch=5), cpu_n=50000)
if "LEV" in which: run("150bp levenshtein score", 2000000, 150, 0.02, 1002, dict(distance="levenshtein", span="end-to-end", scope="score"), cpu_n=100000)
if "LIN" in which: run("150bp gap-linear score", 2000000, 150, 0.02, 1002, dict(distance="linear", span="end-to-end", scope="score"), cpu_n=100000)
if "LEVf" in which: run("150bp levenshtein full", 2000000, 150, 0.02, 1002, dict(distance="levenshtein", span="end-to-end", scope="full"), cpu_n=50000)
if "LINf" in which: run("150bp gap-linear full", 2000000, 150, 0.02, 1002, dict(distance="linear", span="end-to-end", scope="full"), cpu_n=50000)
if "INDf" in which: run("150bp indel full", 2000000, 150, 0.02, 1002, dict(distance="indel", span="end-to-end", scope="full"), cpu_n=50000)
if "X100kb" in which: run("100kb exact score, 1024 pairs", 1024, 100000, 0.08, 1005, dict(span="end-to-end", scope="score"), cpu_n=1, reps=1)
if "MN1" in which: run("150bp match=-1 score", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score", match=-1), cpu_n=100000)
if "MN1f" in which: run("150bp match=-1 full", 1000000, 150, 0.02, 1002, dict(scope="full", match=-1), cpu_n=50000)
if "C4am5" in which: run("C4 adaptive mismatch=5 full", 20000, 10000, 0.08, 1004, dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive", mismatch=5), cpu_n=50)
if "C3m5" in which: run("10kb adaptive mismatch=5 full", 50000, 10000, 0.08, 1003, dict(span="end-to-end", scope="full", heuristic="adaptive", mismatch=5), cpu_n=100)
if "C3xm5" in which: run("10kb exact mismatch=5 score (tile kernel)", 4096, 10000, 0.08, 1003, dict(span="end-to-end", scope="score", mismatch=5), cpu_n=16, reps=1)
if "BH10k" in which: run("10kb BiWFA wf-adaptive full", 2000, 10000, 0.08, 1003, dict(span="end-to-end", scope="full", memory_mode="biwfa", heuristic="adaptive"), cpu_n=40)
if "C5a" in which: run("100kb adaptive full", 500, 100000, 0.08, 1005, dict(span="end-to-end", scope="full", heuristic="adaptive"), cpu_n=10)
