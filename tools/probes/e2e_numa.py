"""Development aid: the PCIe-inclusive rate of wfa_hip_align_batch on the C2 batch, by where the caller's pages
live (first-touched on the GPU's NUMA node or on another one) x how the upload workers are bound (WFA_HIP_NUMA = 0 never, 1 always to
the GPU's node, auto = only when the input lives there).  Prints every call's time, the median and the minimum."""
import sys, time, os, glob
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from pywfa_amd import _native, datagen


def cpulist(path):
    out = []
    for part in open(path).read().strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 7
nodes = {int(os.path.basename(p)[4:]): cpulist(p + "/cpulist") for p in glob.glob("/sys/devices/system/node/node[0-9]*")}
allowed = os.sched_getaffinity(0)
print("nodes:", {k: len(v) for k, v in nodes.items()}, "allowed cpus:", len(allowed), flush=True)
b = datagen.generate(n, 150, 0.02, 1002)
cfg = _native.default_config(); cfg.span = 0; cfg.scope = 0
al = _native.Aligner(cfg)
out = (np.zeros(n, np.int32), np.zeros(n, np.int32))
al.align_batch(b, False, out=out)
info = al.upload_info()
print("upload_info (default mode):", info, flush=True)
al.close()
gpu_node = info["gpu_numa_node"]
cands = [gpu_node] + [k for k in sorted(nodes) if k != gpu_node and set(nodes[k]) & allowed][:1] if gpu_node >= 0 else [None]
for src in cands:
    bb = dict(b)
    if src is not None:
        os.sched_setaffinity(0, set(nodes[src]) & allowed)
        bb["seqs"] = np.empty_like(b["seqs"]); bb["seqs"][:] = b["seqs"]          # first touch from a CPU of node `src`
        for k in ("p_off", "p_len", "t_off", "t_len"):
            bb[k] = b[k].copy()
        o2 = (np.zeros(n, np.int32), np.zeros(n, np.int32))
        os.sched_setaffinity(0, allowed)
    else:
        o2 = out
    for mode in ("0", "1", "auto"):
        os.environ["WFA_HIP_NUMA"] = mode
        a = _native.Aligner(cfg)
        ts = []
        for i in range(calls + 1):
            t = time.perf_counter(); s, st, _ = a.align_batch(bb, False, out=o2); ts.append((time.perf_counter() - t) * 1e3)
        inf = a.upload_info()
        a.close()
        ts = ts[1:]
        print(f"input on node {src} (gpu node {gpu_node}) WFA_HIP_NUMA={mode:4s}: median {np.median(ts):6.1f} ms = {n / np.median(ts) / 1e3:6.1f} M aln/s, "
              f"min {min(ts):6.1f} ms, max {max(ts):6.1f} ms; input node seen {inf['input_numa_node']}, bound {inf['workers_bound']}, "
              f"threads {inf['pack_threads']}+{inf['copy_threads']}  calls: {' '.join(f'{x:.0f}' for x in ts)}", flush=True)
        assert len(os.sched_getaffinity(0)) == len(allowed), "the caller's thread was re-bound"
print("mean score", float(s.mean()))
