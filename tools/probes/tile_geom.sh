#!/bin/bash
# geometry of the tile stage on mid-length exact reads (development aid)
for c in "1000 0.05 150000 score" "600 0.10 250000 score" "2000 0.01 75000 score" "1000 0.05 150000 full" "2000 0.05 75000 full"; do
  for env in "X=1" "WFA_HIP_TILE_WT=64 WFA_HIP_TILE_T=4" "WFA_HIP_TILE_WT=128 WFA_HIP_TILE_T=4" "WFA_HIP_TILE_WT=128 WFA_HIP_TILE_T=12" "WFA_HIP_TILE_WT=192 WFA_HIP_TILE_T=8" "WFA_HIP_TILE_THREADS=256" "WFA_HIP_TILE_PER_CU=8" "WFA_HIP_TILE_PER_CU=32" "WFA_HIP_NO_BAND=1"; do
    echo -n "$env :: "; env $env python tools/probes/stage_probe.py $c 2>/dev/null | tail -1
  done
done
