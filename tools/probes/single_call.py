"""Per-call latency of wavefront_align (one pair per call) — development aid.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pywfa_amd
from pywfa_amd import datagen

b = datagen.generate(64, 150, 0.02, 5)
pairs = [datagen.pair_strings(b, i) for i in range(64)]
for scope in ("score", "full"):
    a = pywfa_amd.WavefrontAligner(pairs[0][0], span="end-to-end", scope=scope)
    for _ in range(200):
        a.wavefront_align(pairs[1][1], pairs[1][0])
    n = int(os.environ.get("N_CALLS", "2000"))
    t0 = time.perf_counter()
    for i in range(n):
        a.wavefront_align(pairs[i & 63][1], pairs[i & 63][0])
    dt = (time.perf_counter() - t0) / n
    # the C-ABI call alone
    nat = a._native
    pb = [p[0].encode() for p in pairs]; tb = [p[1].encode() for p in pairs]
    t0 = time.perf_counter()
    for i in range(n):
        nat.align_pair(pb[i & 63], tb[i & 63], scope == "full")
    dc = (time.perf_counter() - t0) / n
    print(f"scope={scope}: wavefront_align {dt * 1e6:.2f} us per call, align_pair {dc * 1e6:.2f} us  (inline={'off' if os.environ.get('WFA_HIP_NO_TINY_INLINE') else 'on'}, poll={'off' if os.environ.get('WFA_HIP_NO_TINY_POLL') else 'on'})")
    a.close()
