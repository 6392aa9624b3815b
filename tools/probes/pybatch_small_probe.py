"""Development aid: wavefront_align_batch(list of str) for small lists."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pywfa_amd import WavefrontAligner, datagen
b = datagen.generate(2048, 150, 0.02, 1001)
P, T = [], []
for i in range(2048):
    p, t = datagen.pair_strings(b, i); P.append(p); T.append(t)
for scope in ("score", "full"):
    a = WavefrontAligner(scope=scope, span="end-to-end")
    for n in (1, 10, 100, 1000, 2048):
        a.wavefront_align_batch(T[:n], P[:n])
        t0 = time.time()
        for _ in range(20): out = a.wavefront_align_batch(T[:n], P[:n])
        dt = (time.time() - t0) / 20
        print(f"scope={scope} n={n:5d}: {dt * 1e6:8.1f} us per call = {dt / n * 1e6:6.2f} us per pair", flush=True)
    if scope == "full": print(out["cigarstrings"][0], len(out["cigarstrings"]))
