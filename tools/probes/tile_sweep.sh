# geometry sweep of the tiled wide-wavefront kernel (T, tile width, threads per workgroup): tile_sweep.sh <gpu_perf case> "<T Wt threads>" ...
export BRIEF=1 WFA_HIP_STAGE_TIMING=1
which=$1; shift
for geo in "$@"; do
  set -- $geo
  echo "== T=$1 Wt=$2 threads=$3"
  WFA_HIP_TILE_T=$1 WFA_HIP_TILE_WT=$2 WFA_HIP_TILE_THREADS=$3 timeout 300 python tools/gpu_perf.py $which 2>&1 | grep "tile stage\|tile profile\|mism" | tail -3
done
