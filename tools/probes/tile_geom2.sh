#!/bin/bash
# tile geometry between 3 and 8 kb (development aid)
for c in "3000 0.05 50000 score" "5000 0.05 30000 score" "7000 0.05 20000 score" "5000 0.05 30000 full" "10000 0.08 8192 score"; do
  for env in "X=1" "WFA_HIP_TILE_THREADS=128" "WFA_HIP_TILE_THREADS=128 WFA_HIP_TILE_WT=128 WFA_HIP_TILE_T=8" "WFA_HIP_TILE_THREADS=256 WFA_HIP_TILE_WT=128 WFA_HIP_TILE_T=8" "WFA_HIP_TILE_THREADS=128 WFA_HIP_TILE_WT=256 WFA_HIP_TILE_T=8" "WFA_HIP_TILE_THREADS=192"; do
    echo -n "$env :: "; env $env python tools/probes/stage_probe.py $c 2>/dev/null | tail -1
  done
done
echo "== E10 stages"; WFA_HIP_STAGE_TIMING=1 NO_CPU=1 BRIEF=1 python tools/gpu_perf.py E10 2>&1 | grep -v "^\[wfa_hip\] *$" | tail -14 | cut -c1-250
