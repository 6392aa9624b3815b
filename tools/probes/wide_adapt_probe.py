import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import common
from pywfa_amd import datagen, _native
batch = datagen.generate(160, 30000, 0.12, 9400)
oc, nc = common.configs_pair(span="end-to-end", scope="full", heuristic="adaptive", max_distance_threshold=400)
al = _native.Aligner(nc); rb = al.batch(batch); rb.run(); rb.sync(); print("handed to general:", rb.fallback_pairs()); rb.close(); al.close()
