"""Development aid: wall time of consecutive runs of one small resident batch of long reads (is anything initialised lazily after the first run?)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import common
from pywfa_amd import datagen, _native
for prev in (False, True):
    if prev:   # another aligner used and closed just before, as tools/gpu_perf.py does between configurations
        b0 = datagen.generate(8, 10000, 0.08, 1003)
        oc, nc = common.configs_pair(span="end-to-end", scope="full")
        al0 = _native.Aligner(nc); rb0 = al0.batch(b0); rb0.run(); rb0.sync(); rb0.results(True); rb0.close(); al0.close()
    batch = datagen.generate(8, 10000, 0.08, 1003)
    oc, nc = common.configs_pair(span="end-to-end", scope="full", heuristic="adaptive")
    al = _native.Aligner(nc); rb = al.batch(batch)
    for i in range(4):
        t0 = time.time(); rb.run(); rb.sync(); print("prev" if prev else "first", "run", i, round((time.time() - t0) * 1e3, 2), "ms", flush=True)
    rb.close(); al.close()
