#!/usr/bin/env python3
"""Latency of single-pair calls through the Python surface and through the C ABI (development aid)."""
import time, sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from pywfa_amd import WavefrontAligner, _native, datagen
import common
random.seed(1)
def mut(s):
    s = list(s)
    for i in range(3): s[random.randrange(len(s))] = random.choice("ACGT")
    return "".join(s)
p = "".join(random.choice("ACGT") for _ in range(150)); t = mut(p)
for kw in (dict(), dict(scope="score", span="end-to-end")):
    a = WavefrontAligner(p, **kw)
    a.wavefront_align(t)
    t0 = time.perf_counter()
    for _ in range(300): a.wavefront_align(t)
    dt = (time.perf_counter() - t0) / 300
    print(kw, "python surface per call us %.1f" % (dt * 1e6), a.score)
    _, nc = common.configs_pair(**kw)
    al = _native.Aligner(nc); batch = datagen.from_strings([p], [t])
    full = kw.get("scope", "full") == "full"
    al.align_batch(batch, full)
    t0 = time.perf_counter()
    for _ in range(300): al.align_batch(batch, full)
    dt = (time.perf_counter() - t0) / 300
    print(kw, "C ABI align_batch (via ctypes) per call us %.1f" % (dt * 1e6))
    if os.environ.get("WFA_HIP_TIMING"):
        al.align_batch(batch, full)
    al.close()
# the one-pair entry (wfa_hip_align_pair) through ctypes
for kw in (dict(), dict(scope="score", span="end-to-end")):
    _, nc = common.configs_pair(**kw)
    al = _native.Aligner(nc)
    full = kw.get("scope", "full") == "full"
    pb, tb = p.encode(), t.encode()
    al.align_pair(pb, tb, full)
    t0 = time.perf_counter()
    for _ in range(2000): al.align_pair(pb, tb, full)
    print(kw, "C ABI align_pair (via ctypes) per call us %.1f" % ((time.perf_counter() - t0) / 2000 * 1e6))
    al.close()
