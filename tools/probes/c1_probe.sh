export NO_CPU=1 BRIEF=1 TMPDIR=/tmp
O=gpurun_out/r3i
run() { echo "== $*" >> $O.log; env "$@" python tools/gpu_perf.py $CFG >> $O.log 2>&1; }
rm -f $O.log
CFG=C3x
for w2 in 0 1; do for th in 128 256 384 512; do run WFA_HIP_WIDE2=$w2 WFA_HIP_WIDE_GROWS=1 WFA_HIP_WIDE_THREADS=$th; done; done
CFG=C3xf
for w2 in 0 1; do for th in 256 512; do run WFA_HIP_WIDE2=$w2 WFA_HIP_WIDE_GROWS=1 WFA_HIP_WIDE_THREADS=$th; done; done
CFG=C4xs
for th in 256 512 1024; do run WFA_HIP_WIDE_THREADS=$th; done
CFG=C4x
for th in 512 1024; do run WFA_HIP_WIDE_THREADS=$th; done
cat $O.log
