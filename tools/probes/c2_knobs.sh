#!/bin/bash
# C2 under the lane kernel's launch knobs (development aid)
cd "$(dirname "$0")/../.."
for v in "" "WFA_HIP_LANE_WAVES_PER_CU=32" "WFA_HIP_LANE_WAVES_PER_CU=40" "WFA_HIP_LANE_WAVES_PER_CU=56" "WFA_HIP_LANE_WAVES_PER_CU=64" "WFA_HIP_LANE_WAVES_PER_CU=96" "WFA_HIP_LANE_REFILL_MIN=4" "WFA_HIP_LANE_REFILL_MIN=12" "WFA_HIP_LANE_REFILL_MIN=16" "WFA_HIP_LANE_REFILL_MIN=24" "WFA_HIP_FAST_WAVES_PER_CU=128" "WFA_HIP_FAST_WAVES_PER_CU=512"; do
  echo -n "$v: "; env $v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
done
