import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pywfa_amd import datagen, _native
import common
for (n, L) in ((8, 150), (400, 100), (6, 10000), (40, 1500)):
    batch = datagen.generate(n, L, 0.08, 3)
    for kw in (dict(span="end-to-end"), dict(distance="affine2p", span="end-to-end")):
        oc, nc = common.configs_pair(scope="full", memory_mode="biwfa", **kw)
        t0 = time.time()
        for _ in range(5):
            common.gpu_run(nc, batch, True, resident=True)
        t1 = time.time()
        al = _native.Aligner(nc); rb = al.batch(batch); rb.run(); rb.sync()
        t2 = time.time()
        for _ in range(5): rb.run(); rb.sync()
        t3 = time.time()
        rb.close(); al.close()
        print(f"n={n} L={L} {kw}: gpu_run {(t1 - t0) / 5 * 1e3:.1f} ms per call; first run {(t2 - t1) * 1e3:.1f} ms; later runs {(t3 - t2) / 5 * 1e3:.2f} ms", flush=True)
