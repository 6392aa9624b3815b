#!/usr/bin/env python3
"""C4-shaped probes of the banded gap-affine-2p kernel (development aid)."""
import sys, os
sys.argv = ["x", "none"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["BRIEF"] = "1"
import gpu_perf
kw = dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, heuristic="adaptive")
n = int(os.environ.get("N", "8192"))
gpu_perf.run(f"C4a {os.environ.get('SCOPE', 'full')} n={n}", n, 10000, 0.08, 1004, dict(kw, scope=os.environ.get("SCOPE", "full")), cpu_n=20, reps=1)
