import os, sys
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rtc_probe.py")).read().split("S = dict(span=")[0])
S = dict(span="end-to-end", scope="score")
run("150bp affine match=-1 score", 2_000_000, 150, 0.02, dict(S, match=-1))
run("150bp affine match=-1 full (pywfa default span)", 1_000_000, 150, 0.02, dict(scope="full", match=-1))
run("150bp affine match=-2 mismatch=5 score", 2_000_000, 150, 0.02, dict(S, match=-2, mismatch=5))
run("150bp 2p match=-2 full", 200_000, 150, 0.02, dict(distance="affine2p", scope="full", match=-2))
run("150bp linear match=-1 score", 2_000_000, 150, 0.02, dict(distance="linear", span="end-to-end", scope="score", match=-1))
run("10kb adaptive full match=-1", 10_000, 10000, 0.08, dict(span="end-to-end", scope="full", heuristic="adaptive", match=-1), cpu_n=50)
run("10kb exact score match=-1 (tile kernel)", 1000, 10000, 0.08, dict(span="end-to-end", scope="score", match=-1), cpu_n=20)
os.environ["WFA_HIP_RTC_ALL"] = "1"
run("RTC_ALL: 150bp score default", 2_000_000, 150, 0.02, dict(S))
run("RTC_ALL: 150bp full default", 1_000_000, 150, 0.02, dict(scope="full"))
run("RTC_ALL: 10kb adaptive full", 20_000, 10000, 0.08, dict(span="end-to-end", scope="full", heuristic="adaptive"), cpu_n=100)
