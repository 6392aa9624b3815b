"""Development aid: wall time of wfa_hip_align_batch (host ASCII in -> host results out) and of the 2-bit entry on the C2 batch."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from pywfa_amd import _native, datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 7
b = datagen.generate(n, 150, 0.02, 1002)
cfg = _native.default_config(); cfg.span = 0; cfg.scope = 0
al = _native.Aligner(cfg)
out = (np.zeros(n, np.int32), np.zeros(n, np.int32))
ts = []
for i in range(calls):
    t = time.time(); s, st, _ = al.align_batch(b, False, out=out); ts.append((time.time() - t) * 1e3); print('e2e ms', ts[-1], flush=True)
print(f"ASCII in: median {np.median(ts[1:]):.1f} ms = {n / np.median(ts[1:]) / 1e3:.1f} M aln/s, min {min(ts):.1f} ms")
print("mean score", float(s.mean()), "nonzero status", int((st != 0).sum()))
pk = datagen.to_packed2bits(b)
out2 = (np.zeros(n, np.int32), np.zeros(n, np.int32))
ts = []
for i in range(calls):
    t = time.time(); s2, st2, _ = al.align_batch(pk, False, out=out2); ts.append((time.time() - t) * 1e3); print('e2e 2-bit ms', ts[-1], flush=True)
print(f"2-bit in: median {np.median(ts[1:]):.1f} ms = {n / np.median(ts[1:]) / 1e3:.1f} M aln/s, min {min(ts):.1f} ms")
assert np.array_equal(s, s2) and np.array_equal(st, st2)
print("upload_info", al.upload_info())
