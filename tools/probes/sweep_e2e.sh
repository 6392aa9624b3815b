for t in 16 32 48 64 96; do echo "PACK_THREADS=$t"; WFA_HIP_NUMA=0 WFA_HIP_PACK_THREADS=$t WFA_HIP_TIMING=1 python tools/probes/e2e_probe.py 2>&1 | grep "host pack\|e2e ms" | tail -6 | tr '\n' ' '; echo; done
for c in 2 4 16; do echo "PIPE_CHUNK=$c"; WFA_HIP_NUMA=0 WFA_HIP_PIPE_CHUNK=$c WFA_HIP_TIMING=1 python tools/probes/e2e_probe.py 2>&1 | grep "host pack\|e2e ms" | tail -6 | tr '\n' ' '; echo; done
echo "UP_STREAMS=1"; WFA_HIP_NUMA=0 WFA_HIP_UP_STREAMS=1 WFA_HIP_TIMING=1 python tools/probes/e2e_probe.py 2>&1 | grep "host pack\|e2e ms" | tail -6 | tr '\n' ' '; echo
