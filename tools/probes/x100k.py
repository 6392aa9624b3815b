"""Development aid: exact 100 kb (score) at several batch sizes, tiled int32 rows on / off."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["BRIEF"] = "1"; os.environ.setdefault("NO_CPU", "1")
sys.argv = ["x", "none"]
import gpu_perf
for n in [int(x) for x in os.environ.get("NS", "64 256 512").split()]:
    gpu_perf.run(f"100kb exact score n={n} TILE32={os.environ.get('WFA_HIP_TILE32', '1')}", n, 100000, 0.08, 1005, dict(span="end-to-end", scope="score"), cpu_n=1, reps=1)
