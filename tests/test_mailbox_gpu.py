"""The resident one-pair kernel (round 6, csrc/wfa_slim.hpp: wfa_slim_kernel_mailbox): pywfa's one-alignment-per-call loop served from a
mailbox in pinned host memory instead of a kernel launch per call.  What these tests pin down beyond tests/test_parity_gpu.py's single-call
tests (which now run through it): instances that come and go (idle time, configuration changes, batches in between, two aligners), the
launch-per-call path as the fall-back, and — the one way this could hurt — a process that exits with an instance still on the device."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import datagen

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _pairs(n, seed, L=150, e=0.04):
    b = datagen.generate(n, L, e, seed)
    return b, [datagen.pair_strings(b, i) for i in range(n)]


def _check(al, oc, full, pairs, expect, idx):
    p, t = pairs[idx]
    score, status, ops = al.align_pair(p.encode(), t.encode(), full)
    assert (score, status) == (int(expect["score"][idx]), int(expect["status"][idx])), idx
    if full:
        assert ops == expect["cigars"][idx], idx


@pytest.mark.parametrize("kw", [dict(scope="score", span="end-to-end"), dict(scope="full"), dict(scope="full", heuristic="adaptive", span="end-to-end"),
                                dict(scope="full", mismatch=4, gap_opening=6, gap_extension=1),
                                dict(scope="score", span="ends-free", pattern_begin_free=5, pattern_end_free=7, text_begin_free=3, text_end_free=2)])
def test_instances_come_and_go(gpu, kw, monkeypatch):
    """Calls in a row (one instance serves them), a pause longer than the instance's idle time (the next call starts a new one), a batch
    in between (it tells the instance to leave), reads of other lengths (the instance's arguments do not depend on them)."""
    from pywfa_amd import _native
    monkeypatch.setenv("WFA_HIP_MAILBOX_IDLE_US", "300")
    batch, pairs = _pairs(48, 9100)
    long_b, long_pairs = _pairs(8, 9101, L=700, e=0.05)
    oc, nc = common.configs_pair(**kw)
    full = oc.scope == 1
    o = loader.run(loader.oracle(), oc, batch, want_cigar=full)
    ol = loader.run(loader.oracle(), oc, long_b, want_cigar=full)
    al = _native.Aligner(nc)
    try:
        for i in range(16):
            _check(al, oc, full, pairs, o, i)
        time.sleep(0.02)                                     # (the instance has left by now)
        for i in range(16, 24):
            _check(al, oc, full, pairs, o, i)
            if i % 3 == 0:
                time.sleep(0.002)
        score, status, _ = al.align_batch(batch, full)       # a batch: the device is the batch's
        assert np.array_equal(score, o["score"]) and np.array_equal(status, o["status"])
        for i in range(24, 32):
            _check(al, oc, full, pairs, o, i)
            _check(al, oc, full, long_pairs, ol, i % 8)
    finally:
        al.close()


def test_configuration_changes_restart_the_instance(gpu):
    """Scope, penalties and heuristics change between calls (pywfa's setters): an instance started for another configuration leaves first."""
    import pywfa_amd
    batch, pairs = _pairs(24, 9200)
    a = pywfa_amd.WavefrontAligner(pairs[0][0], span="end-to-end")
    steps = [dict(scope="score"), dict(scope="full"), dict(scope="full", mismatch_penalty=3), dict(scope="score", mismatch_penalty=4),
             dict(scope="full", gap_opening_penalty=4)]
    state = dict(span="end-to-end", mismatch=4, gap_opening=6, gap_extension=2)
    for r, st in enumerate(steps * 2):
        for k, v in st.items():
            setattr(a, k, v)
        state["scope"] = st["scope"]
        if "mismatch_penalty" in st: state["mismatch"] = st["mismatch_penalty"]
        if "gap_opening_penalty" in st: state["gap_opening"] = st["gap_opening_penalty"]
        oc, _ = common.configs_pair(**state)
        for i in range(3):
            p, t = pairs[(3 * r + i) % 24]
            one = datagen.from_strings([p], [t])
            o = loader.run(loader.oracle(), oc, one, want_cigar=(state["scope"] == "full"))
            assert a.wavefront_align(t, p) == int(o["score"][0]), (r, i, state)
            if state["scope"] == "full":
                assert a.cigarstring == common.rle(o["cigars"][0])
    a.close()


def test_two_aligners_take_turns(gpu):
    from pywfa_amd import _native
    batch, pairs = _pairs(20, 9300)
    oc1, nc1 = common.configs_pair(scope="score", span="end-to-end")
    oc2, nc2 = common.configs_pair(scope="full")
    o1 = loader.run(loader.oracle(), oc1, batch, want_cigar=False)
    o2 = loader.run(loader.oracle(), oc2, batch, want_cigar=True)
    a1, a2 = _native.Aligner(nc1), _native.Aligner(nc2)
    try:
        for i in range(20):
            _check(a1, oc1, False, pairs, o1, i)
            _check(a2, oc2, True, pairs, o2, i)
    finally:
        a1.close(); a2.close()


def test_mailbox_off_is_the_launch_per_call_path(gpu, monkeypatch):
    from pywfa_amd import _native
    monkeypatch.setenv("WFA_HIP_MAILBOX", "0")
    batch, pairs = _pairs(12, 9400)
    oc, nc = common.configs_pair(scope="full")
    o = loader.run(loader.oracle(), oc, batch, want_cigar=True)
    al = _native.Aligner(nc)
    try:
        for i in range(12):
            _check(al, oc, True, pairs, o, i)
    finally:
        al.close()


EXIT_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import pywfa_amd
a = pywfa_amd.WavefrontAligner("TCTTTACTCGCGCGTTGGAGAAATACAATAGT", scope="score", span="end-to-end")
for _ in range(50):
    assert a.wavefront_align("TCTATACTGCGCGTTTGGAGAAATAAAATAGT") == -24
print("calls done", flush=True)
{how}
"""


@pytest.mark.parametrize("how", ["sys.exit(0)", "os._exit(0)", "a.close(); sys.exit(0)", "raise SystemExit(3)"])
def test_a_process_may_exit_with_an_instance_on_the_device(gpu, how):
    """The instance leaves by itself after its idle time; neither a normal interpreter exit (aligner never closed), nor os._exit, nor an
    exception hangs the process or the device: the child ends within seconds and the next child finds the device usable."""
    if how in ("a.close(); sys.exit(0)", "raise SystemExit(3)") and os.environ.get("WFA_TEST_FULL") != "1":
        pytest.skip("sampled on the suite's time budget: a child process each (WFA_TEST_FULL=1 runs the four)")
    env = dict(os.environ, WFA_HIP_MAILBOX_IDLE_US="200000")     # (an instance that would idle for 0.2 s: it is still there when the process ends)
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", EXIT_SCRIPT.format(root=ROOT, how=how)], capture_output=True, text=True, timeout=120, env=env)
    assert "calls done" in out.stdout, out.stdout + out.stderr[-2000:]
    assert out.returncode == (3 if "SystemExit(3)" in how else 0), (out.returncode, out.stderr[-2000:])
    assert time.time() - t0 < 60


@pytest.mark.parametrize("mode", ["1", "auto", "0"])
def test_upload_workers_never_rebind_the_caller(gpu, mode, monkeypatch):
    """round 5 bound the upload workers to the GPU's NUMA node — and with them the CALLER's thread, which works a
    share of the pieces: the application's main thread stayed narrowed for good.  Whatever WFA_HIP_NUMA says, the caller's affinity mask
    is what it was after a pipelined upload (>= 256 k pairs), results are unchanged, and the diagnostics say what was done."""
    from pywfa_amd import _native
    monkeypatch.setenv("WFA_HIP_NUMA", mode)
    before = os.sched_getaffinity(0)
    batch = datagen.generate(300000, 100, 0.02, 9500)
    oc, nc = common.configs_pair(scope="score", span="end-to-end")
    idx = np.arange(0, 300000, 300)
    o = loader.run(loader.oracle(), oc, datagen.subset(batch, idx), want_cigar=False)
    al = _native.Aligner(nc)
    try:
        for _ in range(2):
            score, status, _ = al.align_batch(batch, False)
            assert os.sched_getaffinity(0) == before
        info = al.upload_info()
    finally:
        al.close()
    assert np.array_equal(score[idx], o["score"]) and int((status != 0).sum()) == 0
    assert info["numa_mode"] == {"0": 0, "1": 1, "auto": 2}[mode] or info["gpu_node_cpus"] == 0
    assert info["pack_threads"] >= 1 and info["process_cpus"] == len(before)
    if mode == "0":
        assert info["workers_bound"] == 0
