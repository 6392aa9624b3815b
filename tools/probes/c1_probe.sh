export NO_CPU=1 BRIEF=1 TMPDIR=/tmp
O=gpurun_out/r3k
run() { echo "== $*" >> $O.log; env "$@" python tools/gpu_perf.py $CFG >> $O.log 2>&1; }
rm -f $O.log
CFG=C1
run A=1
run WFA_HIP_LANE_MIN_PAIRS=128
run WFA_HIP_LANE_MIN_PAIRS=64
run WFA_HIP_LANE_REFILL_MIN=4
run WFA_HIP_LANE_REFILL_MIN=12
run WFA_HIP_LANE_REFILL_MIN=16
run WFA_HIP_SEGFULL_STAGES=2
cat $O.log
