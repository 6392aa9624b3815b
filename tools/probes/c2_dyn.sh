#!/bin/bash
# C2 by the lane kernel's run-time slices: chunk size, waves per CU, idle lanes that trigger a refill (development aid)
mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/lane_dyn.log
run() { echo -n "$* : " >> gpurun_out/r05/lane_dyn.log; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])" >> gpurun_out/r05/lane_dyn.log; }
run WFA_HIP_LANE_DYN=0
for c in 64 128 192 256; do run WFA_HIP_LANE_DYN=$c; done
for w in 8 12 20 24 32; do run WFA_HIP_LANE_DYN=128 WFA_HIP_LANE_DYN_WAVES=$w; done
for r in 2 4 6 12 16; do run WFA_HIP_LANE_DYN=128 WFA_HIP_LANE_REFILL_MIN=$r; done
cat gpurun_out/r05/lane_dyn.log
