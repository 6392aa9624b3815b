"""Development aid: bench.py's C5 multi-device leg after other aligners of the process have come and gone / are still alive (does the
leg's rate depend on what the process did before?)."""
import sys, os, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from pywfa_amd import _native, datagen
mode = sys.argv[1] if len(sys.argv) > 1 else "alive"
keep = []
b = datagen.generate(200000, 150, 0.02, 5)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    al = _native.Aligner(bench.native_config(scope="score", span="end-to-end"))
    al.align_batch(b, False)
    if mode == "alive": keep.append(al)
    else: al.close()
r = bench.multi_leg(8192)
print(mode, len(keep), {k: round(v["alignments_per_s"]) for k, v in r.items() if isinstance(v, dict)}, flush=True)
