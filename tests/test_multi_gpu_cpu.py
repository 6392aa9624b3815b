"""The N > 1 path of bench.py on CPU: two processes over gloo (127.0.0.1).  Each rank takes its shard of
the seeded stream exactly as bench.py does (shard_first), the checker aligns it, the step times are
MAX-reduced; the union of the shards must equal the unsharded stream and results must not depend on the
sharding.  (The GPU kernels themselves are covered by the -m gpu tests; pairs are independent, so there is
no collective on the data path to test.)"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

WORKER = textwrap.dedent('''
    import json, os, sys, time
    import numpy as np
    sys.path.insert(0, os.environ["WFA_ROOT"])
    import bench
    from oracle import loader
    from pywfa_amd import datagen
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist = bench.dist_setup("gloo", 0)
    P = 500
    first = bench.shard_first(rank, P)
    batch = datagen.generate(P, 150, 0.02, datagen.SEEDS["C2"], first=first)
    # the product's own shard planner (wfa_hip_plan_shards, host only) on a ragged batch: every rank plans the same
    # shards and aligns its own; rank 0 checks the union below
    from pywfa_amd import _native
    ragged = datagen.from_strings(["ACGT" * (1 + (7 * i) % 40) for i in range(300)], ["ACGA" * (1 + (5 * i) % 33) for i in range(300)])
    sb = _native.plan_shards(ragged["p_len"], ragged["t_len"], world)
    mine = np.arange(sb[rank], sb[rank + 1])
    r_o = loader.run(loader.oracle(), loader.make_config(span="end-to-end", scope="score"), datagen.subset(ragged, mine))
    np.save(os.path.join(os.environ["WFA_OUT"], f"ragged_{rank}.npy"), r_o["score"])
    np.save(os.path.join(os.environ["WFA_OUT"], f"ragged_sb_{rank}.npy"), sb)
    bench.dist_barrier(dist, "gloo")
    t0 = time.perf_counter()
    o = loader.run(loader.oracle(), loader.make_config(span="end-to-end", scope="score"), batch)
    time.sleep(0.05 * (rank + 1))
    bench.dist_barrier(dist, "gloo")
    elapsed = time.perf_counter() - t0
    mx = bench.dist_max(dist, "gloo", elapsed)
    assert mx >= elapsed - 1e-9
    np.save(os.path.join(os.environ["WFA_OUT"], f"score_{rank}.npy"), o["score"])
    with open(os.path.join(os.environ["WFA_OUT"], f"meta_{rank}.json"), "w") as f:
        json.dump({"first": first, "elapsed": elapsed, "max": mx}, f)
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_rank_sharding_over_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), WFA_ROOT=ROOT, WFA_OUT=str(tmp_path), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    import json
    sys.path.insert(0, ROOT)
    from oracle import loader
    from pywfa_amd import datagen
    whole = datagen.generate(1000, 150, 0.02, datagen.SEEDS["C2"])
    o = loader.run(loader.oracle(), loader.make_config(span="end-to-end", scope="score"), whole)
    got = np.concatenate([np.load(tmp_path / f"score_{r}.npy") for r in range(2)])
    assert np.array_equal(got, o["score"])            # shards partition the stream, results unchanged
    # the ragged batch sharded by the product's planner: contiguous, complete, balanced by bases, results unchanged
    ragged = datagen.from_strings(["ACGT" * (1 + (7 * i) % 40) for i in range(300)], ["ACGA" * (1 + (5 * i) % 33) for i in range(300)])
    sb0, sb1 = np.load(tmp_path / "ragged_sb_0.npy"), np.load(tmp_path / "ragged_sb_1.npy")
    assert np.array_equal(sb0, sb1) and sb0[0] == 0 and sb0[-1] == 300 and (np.diff(sb0) >= 0).all()
    work = (ragged["p_len"].astype(np.int64) + ragged["t_len"] + 16)
    halves = [work[sb0[i]:sb0[i + 1]].sum() for i in range(2)]
    assert abs(halves[0] - halves[1]) <= work.max() + 1
    ro = loader.run(loader.oracle(), loader.make_config(span="end-to-end", scope="score"), ragged)
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"ragged_{r}.npy") for r in range(2)]), ro["score"])
    metas = [json.load(open(tmp_path / f"meta_{r}.json")) for r in range(2)]
    assert [m["first"] for m in metas] == [0, 500]
    assert abs(metas[0]["max"] - metas[1]["max"]) < 1e-6 and metas[0]["max"] >= max(m["elapsed"] for m in metas) - 1e-6
