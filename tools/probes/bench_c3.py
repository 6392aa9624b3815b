"""Development aid: bench.py's C3 / C4-adaptive extra configurations alone, in a fresh process or after the C2 + C1 legs (argv[1] = 'after')."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from pywfa_amd import datagen, _native
if len(sys.argv) > 1 and sys.argv[1] == "after":
    cfg = _native.default_config(); cfg.span, cfg.scope = _native.SPAN["end-to-end"], _native.SCOPE["score"]
    batch = datagen.generate(10000000, 150, 0.02, datagen.SEEDS["C2"])
    al = _native.Aligner(cfg, device=0)
    r = bench.run_resident(al, batch, 5, 1, False)
    outs = (np.zeros(10000000, np.int32), np.zeros(10000000, np.int32))
    for _ in range(8): al.align_batch(batch, False, out=outs)
    pk = datagen.to_packed2bits(batch)
    for _ in range(6): al.align_batch(pk, False, out=outs)
    del pk
    al.close()
    os.environ["WFA_HIP_NUMA"] = "1"; alb = _native.Aligner(cfg, device=0); del os.environ["WFA_HIP_NUMA"]
    for _ in range(6): alb.align_batch(batch, False, out=outs)
    alb.close()
    os.environ["WFA_HIP_HOST_PACK"] = "0"; al0 = _native.Aligner(cfg, device=0); del os.environ["WFA_HIP_HOST_PACK"]
    for _ in range(3): al0.align_batch(batch, False, out=outs)
    al0.close(); del batch
    x = bench.extra_config(name="C1", n=1_000_000, length=150, error=0.02, seed=datagen.SEEDS["C1"], cfg_kw=dict(scope="full"), scheme="explicit", survey_bytes=236, cpu_pairs=2000, cpu_budget=1.0)
    print("C1", x["ms_per_step"])
for rep in range(2):
    x = bench.extra_config(name="C3", n=100_000, length=10000, error=0.08, seed=datagen.SEEDS["C3"], cfg_kw=dict(span="end-to-end", scope="full", heuristic="adaptive"), scheme="piggyback", survey_bytes=114e3, cpu_pairs=20, cpu_budget=1.0)
    print("C3 ms/step", x["ms_per_step"], "kernel_ms", x["kernel_ms"], flush=True)
