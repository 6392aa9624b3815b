import sys, os
sys.argv=["x","none"]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"tools"))
os.environ["BRIEF"]="1"
import gpu_perf
kw=dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full", heuristic="adaptive")
for n in (8192, 32768, 65536):
    gpu_perf.run(f"C4a n={n}", n, 10000, 0.08, 1004, kw, cpu_n=50, reps=1)
