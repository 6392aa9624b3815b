#!/bin/bash
# Build libwfa_hip.so (HIP kernels + C ABI, gfx950) and libwfa_synth.so (host generator) in-tree.
set -e
cd "$(dirname "$0")"
OUT=..
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I../../include -I. \
  -Wno-unused-result wfa_hip.hip -o $OUT/libwfa_hip.so ${WFA_HIP_EXTRA_FLAGS}
gcc -O3 -fPIC -fopenmp -shared synth.c -o libwfa_synth.so
echo "built $OUT/libwfa_hip.so csrc/libwfa_synth.so"
