# the pipelined tail (round 6): C3 / C4-adaptive at bench size with and without it
for v in 1 0; do echo "WFA_HIP_PIPE_TAIL=$v"; WFA_HIP_PIPE_TAIL=$v BRIEF=1 python tools/gpu_perf.py C3big C4abig C3 C4a; done
