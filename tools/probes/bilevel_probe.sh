#!/bin/bash
# BiWFA level by level on the GPU: parity, then rates with the per-level timing (development aid).
mkdir -p gpurun_out/bl
rm -f gpurun_out/bl/*.log
export BRIEF=1
( timeout 900 python -m pytest tests/test_biwfa_gpu.py -x -q -m gpu --durations=5 2>&1 | tail -14 ) > gpurun_out/bl/test_biwfa.log 2>&1
( timeout 600 python -m pytest tests/test_slim_gpu.py tests/test_python_surface_gpu.py -x -q -m gpu -k "without_a_heuristic or surface" 2>&1 | tail -5 ) > gpurun_out/bl/test_single.log 2>&1
( WFA_HIP_STAGE_TIMING=1 timeout 600 python tools/gpu_perf.py B10k 2>&1 | grep -i "biwfa" | tail -2 ) > gpurun_out/bl/levels.log 2>&1
( WFA_HIP_STAGE_TIMING=1 timeout 600 python tools/gpu_perf.py B1k 2>&1 | grep -i "biwfa" | tail -2 ) >> gpurun_out/bl/levels.log 2>&1
( timeout 900 python tools/gpu_perf.py B10k B10kbig B1k BH10k B100k 2>&1 | tail -6 ) > gpurun_out/bl/perf.log 2>&1
( WFA_HIP_BILEVEL_NO_LDS=1 timeout 900 python tools/gpu_perf.py B10k B1k 2>&1 | tail -3 ) > gpurun_out/bl/perf_nolds.log 2>&1
( timeout 300 python tools/probes/pybatch_probe.py 2>&1 | tail -8 ) > gpurun_out/bl/pybatch.log 2>&1
tail -n 20 gpurun_out/bl/*.log
