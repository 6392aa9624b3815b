#!/bin/bash
# wfa_slim_kernel on the GPU: parity, then C4-adaptive by the occupancy the 2p form is compiled for.
mkdir -p gpurun_out/slim
rm -f gpurun_out/slim/*.log
export BRIEF=1
( timeout 1500 python -m pytest tests/test_slim_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/slim/test_slim.log 2>&1
( timeout 600 python tools/gpu_perf.py C3 C4a 2>&1 | tail -6 ) > gpurun_out/slim/perf_slim.log 2>&1
for w in 2 3; do
( echo "2p form compiled for $w waves per SIMD"; WFA_HIP_LIB=$PWD/pywfa_amd/libwfa_hip_w$w.so WFA_HIP_STAGE_TIMING=1 NO_CPU=1 timeout 600 python tools/gpu_perf.py C4a 2>&1 | grep -i "stage 0\|C4" | tail -2 ) >> gpurun_out/slim/occupancy2p.log 2>&1
done
( echo "4 waves"; WFA_HIP_STAGE_TIMING=1 NO_CPU=1 timeout 600 python tools/gpu_perf.py C4a 2>&1 | grep -i "stage 0\|C4" | tail -2 ) >> gpurun_out/slim/occupancy2p.log 2>&1
tail -n 20 gpurun_out/slim/*.log
