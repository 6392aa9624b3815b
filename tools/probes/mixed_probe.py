import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import common
from oracle import loader
from pywfa_amd import datagen
rng = np.random.default_rng(3)
pats, txts = [], []
for i in range(600):
    L = int(rng.choice([20, 60, 150, 150, 150, 300, 700, 3000]))
    b = datagen.generate(1, L, float(rng.choice([0.0, 0.02, 0.1])), 10_000 + i)
    p, t = datagen.pair_strings(b, 0)
    if i % 7 == 0:
        p = p[: L // 2] + "N" + p[L // 2 + 1:]
    pats.append(p); txts.append(t)
batch = datagen.from_strings(pats, txts)
for kw in (dict(span="end-to-end", scope="score"), dict(scope="full"), dict(scope="full", heuristic="adaptive")):
    oc, nc = common.configs_pair(**kw)
    t0 = time.time(); o = loader.run(loader.oracle(), oc, batch); t1 = time.time()
    print(kw, "oracle", round(t1 - t0, 2), flush=True)
    for resident in (False, True):
        t0 = time.time(); score, status, cigars = common.gpu_run(nc, batch, oc.scope == 1, resident); print("  gpu resident", resident, round(time.time() - t0, 2), flush=True)
