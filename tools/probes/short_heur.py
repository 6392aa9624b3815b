"""Development aid: the short-read configurations VERDICT r05 item 5 lists, with the stage timings of the cascade."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["BRIEF"] = "1"; os.environ.setdefault("NO_CPU", "1")
sys.argv = ["x", "none"]
import gpu_perf
which = os.environ.get("WHICH", "adapt xdrop match1 levfull wild").split()
n = int(os.environ.get("N", "2000000"))
if "adapt" in which: gpu_perf.run("150bp 2% wf-adaptive score", n, 150, 0.02, 1002, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=1000)
if "xdrop" in which: gpu_perf.run("150bp 2% X-drop(100) score", n, 150, 0.02, 1002, dict(span="end-to-end", scope="score", heuristic="X-drop", xdrop=100), cpu_n=1000)
if "match1" in which: gpu_perf.run("150bp 2% match=-1 score", n, 150, 0.02, 1002, dict(span="end-to-end", scope="score", match=-1), cpu_n=1000)
if "levfull" in which: gpu_perf.run("150bp 2% levenshtein full", n, 150, 0.02, 1002, dict(span="end-to-end", scope="full", distance="levenshtein"), cpu_n=1000)
if "wild" in which:
    import numpy as np
    from pywfa_amd import datagen
    # every pair holds one N (pattern or text): aligned on bytes with the wildcard rule
    b = datagen.generate(min(n, 500000), 150, 0.02, 1002)
    seqs = b["seqs"].copy()
    pos = (b["p_off"] + 70).astype(np.int64); seqs[pos] = ord("N")
    b2 = dict(b, seqs=seqs)
    gpu_perf.datagen.generate = lambda *a, **k: b2
    gpu_perf.run("150bp 2% wildcard=N, every pair holds an N, score", len(b2["p_len"]), 150, 0.02, 1002, dict(span="end-to-end", scope="score", wildcard="N"), cpu_n=1000)
