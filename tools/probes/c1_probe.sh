export BRIEF=1 NO_CPU=1
python - <<'P'
import os, sys, time
sys.path.insert(0, "tools"); sys.argv = ["x", "none"]
import gpu_perf
for e in (0.005, 0.01, 0.02):
    for h in ("1", "0"):
        os.environ["WFA_HIP_LANE_HEUR"] = h
        gpu_perf.run(f"150bp adaptive score e={e} LANE_HEUR={h}", 2000000, 150, e, 1002, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=1000)
        gpu_perf.run(f"150bp ends-free(8,7,3,2) e={e} LANE_HEUR={h}", 2000000, 150, e, 1002, dict(span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2, scope="score"), cpu_n=1000)
del os.environ["WFA_HIP_LANE_HEUR"]
gpu_perf.run("150bp adaptive score e=0.005 pilot", 2000000, 150, 0.005, 1002, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=1000)
gpu_perf.run("150bp adaptive score e=0.02 pilot", 2000000, 150, 0.02, 1002, dict(span="end-to-end", scope="score", heuristic="adaptive"), cpu_n=1000)
P
