"""Development aid: wall time of wfa_hip_align_batch for small batches of 150 bp pairs."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from pywfa_amd import _native, datagen
b = datagen.generate(16384, 150, 0.02, 1002)
for scope in (0, 1):
    cfg = _native.default_config(); cfg.span = 0; cfg.scope = scope
    al = _native.Aligner(cfg)
    for n in (1, 16, 128, 512, 1024, 2048, 4096, 8192, 8193, 16384):
        sub = datagen.subset(b, np.arange(n))
        al.align_batch(sub, bool(scope))
        t0 = time.time()
        for _ in range(20): al.align_batch(sub, bool(scope))
        dt = (time.time() - t0) / 20
        print(f"scope={'full' if scope else 'score'} n={n:5d}: {dt * 1e6:8.1f} us per call = {dt / n * 1e6:7.2f} us per pair", flush=True)
    al.close()
