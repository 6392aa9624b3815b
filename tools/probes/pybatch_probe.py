"""Development aid: what a Python caller of wavefront_align_batch(list of str) pays per pair."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pywfa_amd import WavefrontAligner, datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
b = datagen.generate(n, 150, 0.02, 1001)
P, T = [], []
for i in range(n):
    p, t = datagen.pair_strings(b, i); P.append(p); T.append(t)
for scope in ("score", "full"):
    a = WavefrontAligner(scope=scope, span="end-to-end")
    a.wavefront_align_batch(T[:1000], P[:1000])
    t0 = time.time(); out = a.wavefront_align_batch(T, P); t1 = time.time()
    msg = f"scope={scope}: {n} pairs of str -> results in {(t1 - t0) * 1e3:.0f} ms = {(t1 - t0) / n * 1e6:.2f} us per pair"
    if scope == "full":
        t2 = time.time(); cs = [out["cigarstrings"][i] for i in range(0, n, 10)]; t3 = time.time()
        msg += f"; reading every 10th CIGAR string: {(t3 - t2) / len(cs) * 1e6:.2f} us each ({cs[0]})"
    print(msg, flush=True)
