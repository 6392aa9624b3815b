"""The driver-facing bench paths on the GPU box: the N > 1 launch (one rank per GPU under torch.distributed.run) and the compact line.
An 8-GPU node is not available to the GPU suite, so two ranks share device 0 and talk over gloo (WFA_BENCH_BACKEND / WFA_BENCH_SHARE_DEVICE,
development aids of bench.py): every line of the N > 1 path runs except RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strict(line):
    def bad(x):
        raise ValueError(x)
    return json.loads(line, parse_constant=bad)


@pytest.mark.gpu
def test_bench_two_ranks_share_one_device(gpu):
    env = dict(os.environ, WFA_BENCH_BACKEND="gloo", WFA_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--pairs", "200000", "--c3-pairs", "1000"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = p.stdout.strip().splitlines()[-1]
    assert len(line) < 8000
    r = _strict(line)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["errors"] == []
    assert r["value"] > 0 and r["ms_per_step"] > 0
    # both halves of the metric: 150 bp (value) and 10 kb (the C3 leg every rank runs), summed over the ranks
    assert r["config"]["c3_alignments_per_s"] > 0 and r["config"]["c3_invalid_transcripts"] == 0
    assert "roofline" in r and r["roofline"]["bound"] == "hbm"
