#!/bin/bash
# wfa_slim_kernel on the GPU: parity (its own test, the banded tests it now serves), then C3 with and without it, and by occupancy.
mkdir -p gpurun_out/slim
rm -f gpurun_out/slim/*.log
export BRIEF=1
( timeout 1200 python -m pytest tests/test_slim_gpu.py -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/slim/test_slim.log 2>&1
( timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "10kb or banded_kernel_penalty or piggyback_history_gives" 2>&1 | tail -8 ) > gpurun_out/slim/test_parity.log 2>&1
( timeout 600 python tools/gpu_perf.py C3 C3s 2>&1 | tail -4 ) > gpurun_out/slim/perf_slim.log 2>&1
for pad in 1 2 3 5 8; do
( echo "pad $pad KB"; WFA_HIP_SLIM_LDS_PAD_KB=$pad WFA_HIP_STAGE_TIMING=1 NO_CPU=1 timeout 600 python tools/gpu_perf.py C3 2>&1 | grep -i "stage 0\|C3" | tail -2 ) >> gpurun_out/slim/occupancy.log 2>&1
done
tail -n 20 gpurun_out/slim/*.log
