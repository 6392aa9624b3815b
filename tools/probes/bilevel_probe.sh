#!/bin/bash
# BiWFA level by level on the GPU: parity, then rates with the per-level timing (development aid).
mkdir -p gpurun_out/bl
rm -f gpurun_out/bl/*.log
export BRIEF=1
( timeout 900 python -m pytest tests/test_biwfa_gpu.py -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/bl/test_biwfa.log 2>&1
( WFA_HIP_STAGE_TIMING=1 timeout 600 python tools/gpu_perf.py B10k 2>&1 | grep -i "biwfa" | tail -6 ) > gpurun_out/bl/levels.log 2>&1
( timeout 900 python tools/gpu_perf.py B10k B10kbig B1k BH10k B100k 2>&1 | tail -6 ) > gpurun_out/bl/perf.log 2>&1
( WFA_HIP_BILEVEL=0 timeout 900 python tools/gpu_perf.py B10k B1k 2>&1 | tail -3 ) > gpurun_out/bl/perf_dfs.log 2>&1
tail -n 20 gpurun_out/bl/*.log
