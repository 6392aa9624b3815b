"""Development aid: wall time of wfa_hip_align_batch (host ASCII in -> host results out) on the C2 batch."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from pywfa_amd import _native, datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
b = datagen.generate(n, 150, 0.02, 1002)
cfg = _native.default_config(); cfg.span = 0; cfg.scope = 0
al = _native.Aligner(cfg)
out = (np.zeros(n, np.int32), np.zeros(n, np.int32))
for i in range(5):
    t = time.time(); s, st, _ = al.align_batch(b, False, out=out); print('e2e ms', (time.time() - t) * 1e3, flush=True)
print("mean score", float(s.mean()), "nonzero status", int((st != 0).sum()))
