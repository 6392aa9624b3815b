#!/bin/bash
# C3 (100 k x 10 kb wf-adaptive, CIGAR) under launch-geometry knobs (development aid)
cd "$(dirname "$0")/../.."
export NO_CPU=1 BRIEF=1
for v in "" "WFA_HIP_NO_DUAL=1" "WFA_HIP_BAND_WAVES_PER_CU=64" "WFA_HIP_BAND_WAVES_PER_CU=96" "WFA_HIP_BAND_WAVES_PER_CU=192" "WFA_HIP_BAND_WAVES_PER_CU=256" "WFA_HIP_BAND_WAVES_PER_CU=32"; do
  echo "== $v"; env $v python3 tools/gpu_perf.py C3 C4abig 2>&1 | grep "aln/s"
done
