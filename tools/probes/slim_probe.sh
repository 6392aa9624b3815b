#!/bin/bash
# the long-read cascade on the GPU: parity, then C3 / C4-adaptive by stage.
mkdir -p gpurun_out/slim
rm -f gpurun_out/slim/*.log
export BRIEF=1
( timeout 1500 python -m pytest tests/test_slim_gpu.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/slim/test_slim.log 2>&1
( timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "10kb or banded_kernel_penalty or piggyback_history or segmented_full or golden" 2>&1 | tail -5 ) > gpurun_out/slim/test_parity.log 2>&1
( WFA_HIP_STAGE_TIMING=1 timeout 600 python tools/gpu_perf.py C3 2>&1 | grep -i "stage\|C3\|C4" | tail -6 ) > gpurun_out/slim/stages.log 2>&1
( timeout 900 python tools/gpu_perf.py C3 C4a C4abig E10 C1 2>&1 | tail -5 ) > gpurun_out/slim/perf.log 2>&1
tail -n 20 gpurun_out/slim/*.log
