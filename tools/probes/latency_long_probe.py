import sys, time, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/pywfa_amd") else ".")
sys.path.insert(0, ".")
from pywfa_amd import WavefrontAligner, datagen
b = datagen.generate(4, 10000, 0.08, 1003)
for kw in (dict(span="end-to-end", scope="full"), dict(span="end-to-end", scope="full", heuristic="adaptive"), dict(distance="affine2p", span="end-to-end", scope="full")):
    p, t = datagen.pair_strings(b, 0)
    a = WavefrontAligner(p, **kw)
    a.wavefront_align(t)
    t0 = time.time()
    for i in range(5): s = a.wavefront_align(t)
    print(kw, "ms per call", (time.time() - t0) / 5 * 1e3, "score", s, len(a.cigarstring))
