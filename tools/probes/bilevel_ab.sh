#!/bin/bash
# A/B of the level-by-level BiWFA knobs (development aid).
mkdir -p gpurun_out/bl
rm -f gpurun_out/bl/ab_*.log
export BRIEF=1 NO_CPU=1
run() { local name=$1; shift; ( env "$@" timeout 600 python tools/gpu_perf.py B10k B1k 2>&1 | tail -2 ) > gpurun_out/bl/ab_$name.log 2>&1; }
run fly2_seql X=1
run fly2_noseql WFA_HIP_BILEVEL_NO_SEQL=1
run fly4_seql WFA_HIP_LIB=pywfa_amd/libwfa_hip_fly4.so
run fly4_noseql WFA_HIP_LIB=pywfa_amd/libwfa_hip_fly4.so WFA_HIP_BILEVEL_NO_SEQL=1
run fly2_seql_wide0 WFA_HIP_BILEVEL_WIDE_LEVELS=0
run fly2_seql_wide2 WFA_HIP_BILEVEL_WIDE_LEVELS=2
run fly2_seql_percu8 WFA_HIP_BILEVEL_PER_CU=8
run fly2_seql_percu32 WFA_HIP_BILEVEL_PER_CU=32
( WFA_HIP_STAGE_TIMING=1 timeout 600 python tools/gpu_perf.py B10k 2>&1 | grep -i "biwfa" | tail -2 ) > gpurun_out/bl/ab_levels.log 2>&1
tail -n 5 gpurun_out/bl/ab_*.log
