#!/bin/bash
# Every rocprofv3 pass behind profiles/ (development aid; run on the GPU box through gpurun):
#   bash tools/gpu_profiles_all.sh <tag>          -> gpurun_out/<tag>_*
# C2 (bench.py --steps 2): issue-rate microbenchmark, SQ / FETCH_SIZE / WRITE_SIZE passes, kernel stats; the region counters of the
# lane kernel (counting build pywfa_amd/libwfa_hip_dbg.so, built beforehand: WFA_BUILD_SUFFIX=_dbg WFA_HIP_EXTRA_FLAGS=-DWFA_LANE_DEBUG_COUNTERS=1
# bash pywfa_amd/csrc/build.sh); then the same passes for the other configurations (tools/gpu_profile_config.sh).
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_counters.sh $TAG
if [ -f pywfa_amd/libwfa_hip_dbg.so ]; then
  WFA_HIP_LIB=pywfa_amd/libwfa_hip_dbg.so WFA_HIP_LANE_DEBUG=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > /dev/null 2> $OUT/${TAG}_lane_regions_raw.txt
  grep -m1 "lane kernel:" $OUT/${TAG}_lane_regions_raw.txt > $OUT/${TAG}_lane_regions.txt
fi
for cfg in C1 C3 C4abig C4x4k C3x8k C3xf8k B10k; do bash tools/gpu_profile_config.sh $TAG $cfg; done
ls $OUT | grep "^${TAG}_" | head -80
