#!/bin/bash
# Build libwfa_hip.so (HIP kernels + C ABI, gfx950) and libwfa_synth.so (host generator) in-tree.
# The kernels are split over translation units (one per kernel family and penalty shape) compiled in parallel.
set -e
cd "$(dirname "$0")"
OUT=..
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result ${WFA_HIP_EXTRA_FLAGS}"
SUF=${WFA_BUILD_SUFFIX:-}
OBJ=build_obj$SUF
mkdir -p $OBJ
JOBS=${WFA_BUILD_JOBS:-$(nproc)}
pids=()
run() {  # bounded parallelism
  while [ "$(jobs -rp | wc -l)" -ge "$JOBS" ]; do sleep 0.2; done
  "$@" &
  pids+=($!)
}
# (the slowest units first)
run $HIPCC $FLAGS -DWFA_TU_INDEX=4 -c k_band.hip -o $OBJ/k_band_4.o
for i in 0 1 2 3 5; do run $HIPCC $FLAGS -DWFA_TU_INDEX=$i -c k_band.hip -o $OBJ/k_band_$i.o; done
for i in 0 1 2 3 4 5 6; do run $HIPCC $FLAGS -DWFA_TU_INDEX=$i -c k_seg.hip -o $OBJ/k_seg_$i.o; done
for i in 0 1 2; do run $HIPCC $FLAGS -DWFA_TU_INDEX=$i -c k_general.hip -o $OBJ/k_general_$i.o; done
for i in 0 1 2 3 4 5 6; do run $HIPCC $FLAGS -DWFA_TU_INDEX=$i -c k_lane.hip -o $OBJ/k_lane_$i.o; done
for i in 0 1 2; do run $HIPCC $FLAGS -DWFA_TU_INDEX=$i -c k_biwfa.hip -o $OBJ/k_biwfa_$i.o; done
for f in k_*.hip; do
  case $f in k_band.hip|k_seg.hip|k_general.hip|k_lane.hip|k_biwfa.hip) ;; *) run $HIPCC $FLAGS -c $f -o $OBJ/${f%.hip}.o ;; esac
done
run $HIPCC $FLAGS -c wfa_hip.hip -o $OBJ/wfa_hip.o
run g++ -O3 -std=c++17 -fPIC -c host_pack.cpp -o $OBJ/host_pack.o   # host code only (AVX-512 / AVX2 / plain C, chosen at run time)
run g++ -O3 -std=c++17 -fPIC -I../../include -c host_cigar.cpp -o $OBJ/host_cigar.o   # host code only (text helpers)
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
[ $fail -eq 0 ] || { echo "build failed"; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $OUT/libwfa_hip$SUF.so -lpthread
gcc -O3 -fPIC -fopenmp -shared synth.c -o libwfa_synth.so
echo "built $OUT/libwfa_hip$SUF.so csrc/libwfa_synth.so"
