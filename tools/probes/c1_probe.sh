export NO_CPU=1 BRIEF=1 TMPDIR=/tmp
O=gpurun_out/r3f
run() { echo "== $*" >> $O.log; env "$@" python tools/gpu_perf.py $CFG >> $O.log 2>&1; }
rm -f $O.log
CFG=N150
run A=1
run WFA_HIP_LANE_LDS_PAD_KB=5
run WFA_HIP_LANE_LDS_PAD_KB=8
run WFA_HIP_LANE_LDS_PAD_KB=12
run WFA_HIP_LANE_LDS_PAD_KB=20
CFG=C1
run A=1
run WFA_HIP_LANE_DEBUG=16
run WFA_HIP_LANE_DEBUG=32
run WFA_HIP_LANE_DEBUG=48
cat $O.log
timeout 600 python -m pytest tests/test_wide_gpu.py -m gpu -x -q -k "beyond_16kb" 2>&1 | tail -5
BRIEF=1 NO_CPU= timeout 900 python tools/gpu_perf.py X30k X30kf X100k 2>&1 | tail -5
