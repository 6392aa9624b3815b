export BRIEF=1 WFA_HIP_STAGE_TIMING=1
for t in 1 0; do echo "== WFA_HIP_TILE=$t"; WFA_HIP_TILE=$t timeout 300 python tools/gpu_perf.py C3x8k C3xf8k C4x4k 2>&1 | grep -v "^$" | tail -20; done
