"""Development aid: what a Python caller of wavefront_align_batch(list of str) pays per pair — by part (the compiled host's
marshalling, the library call) and as a whole, first call (pins the upload ring, sizes the workspace) and steady state."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pywfa_amd import WavefrontAligner, datagen, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
b = datagen.generate(n, 150, 0.02, 1001)
P, T = [], []
for i in range(n):
    p, t = datagen.pair_strings(b, i); P.append(p); T.append(t)
host = _native.compiled_host()
print("compiled host:", host is not None, flush=True)
if host is not None:
    for r in range(3):
        t0 = time.time(); d = host.from_strings(P, T); t1 = time.time()
        print(f"  host.from_strings (patterns + texts): {(t1 - t0) * 1e3:.1f} ms = {n / (t1 - t0) / 1e6:.1f} M pairs/s", flush=True)
    t0 = time.time(); d1 = host.from_strings(P[0], T); t1 = time.time()
    print(f"  host.from_strings (one pattern): {(t1 - t0) * 1e3:.1f} ms", flush=True)
t0 = time.time(); d0 = datagen.from_strings(P, T); t1 = time.time()
print(f"  datagen.from_strings (Python): {(t1 - t0) * 1e3:.1f} ms", flush=True)
for scope in ("score", "full"):
    a = WavefrontAligner(scope=scope, span="end-to-end")
    a.wavefront_align_batch(T[:1000], P[:1000])
    for r in range(4):
        t0 = time.time(); out = a.wavefront_align_batch(T, P); t1 = time.time()
        print(f"scope={scope} call {r}: {n} pairs of str -> results in {(t1 - t0) * 1e3:.0f} ms = {n / (t1 - t0) / 1e6:.1f} M pairs/s", flush=True)
    for r in range(2):
        t0 = time.time(); out2 = a.align_batch(d0); t1 = time.time()
        print(f"  align_batch(prepared batch) alone: {(t1 - t0) * 1e3:.0f} ms", flush=True)
    if scope == "full":
        t2 = time.time(); cs = [out["cigarstrings"][i] for i in range(0, n, 10)]; t3 = time.time()
        print(f"  reading every 10th CIGAR string: {(t3 - t2) / len(cs) * 1e6:.2f} us each ({cs[0]})", flush=True)
