#!/bin/bash
# tile geometry for C4 as written (exact gap-affine-2p, 10 kb, full CIGAR) — development aid
export NO_CPU=1 BRIEF=1
for env in "X=1" "WFA_HIP_TILE_T=12" "WFA_HIP_TILE_T=16" "WFA_HIP_TILE_WT=192" "WFA_HIP_TILE_WT=256 WFA_HIP_TILE_T=16" "WFA_HIP_TILE_WT=256 WFA_HIP_TILE_T=8" "WFA_HIP_TILE_THREADS=128" "WFA_HIP_TILE_THREADS=512" "WFA_HIP_TILE_PER_CU=2" "WFA_HIP_TILE_PER_CU=4" "WFA_HIP_TILE_PER_CU=8"; do
  echo -n "$env :: "; env $env python tools/gpu_perf.py C4x4k 2>&1 | tail -1
done
