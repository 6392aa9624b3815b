import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import common
from pywfa_amd import datagen, _native
os.environ["WFA_HIP_STAGE_TIMING"]="1"
oc, nc = common.configs_pair(span="end-to-end", scope="score")
al = _native.Aligner(nc)
batch = datagen.generate(65536, 150, 0.02, 5)
rb = al.batch(batch); rb.run(); rb.sync()
print("---- second run", file=sys.stderr, flush=True)
rb.run(); rb.sync()
