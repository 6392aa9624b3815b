"""A bounded, fixed-seed slice of tools/gpu_fuzz.py under -m gpu: random configurations (penalties with and
without an instantiation in the library, heuristics, free ends, memory modes incl. BiWFA, step limits, match < 0, one-component
distances), random read lengths / divergences / ragged batches — every pair against the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [20261003, 77])
def test_fuzz_slice(gpu, seed):
    # (the GPU suite runs on a budget — up to 14 rounds, none started after 14 s: on a fresh box the first
    # run-time shapes of a seed compile for seconds each; WFA_TEST_FULL=1 lifts the limit)
    seconds = "0" if os.environ.get("WFA_TEST_FULL") == "1" else "10"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "14", str(seed), seconds], capture_output=True, text=True, timeout=1500)
    tail = "\n".join(out.stdout.splitlines()[-20:])
    assert out.returncode == 0 and "TOTAL BAD 0" in out.stdout, tail + out.stderr[-2000:]
    # (a slice that stopped after one round on a cold run-time-shape cache would still say "TOTAL BAD 0")
    rounds = [int(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("ROUNDS ")]
    assert rounds and rounds[-1] >= 4, tail


@pytest.mark.gpu
@pytest.mark.parametrize("scope", ["score", "full"])
def test_a_run_time_shape_that_cannot_be_built_still_runs(gpu, scope, monkeypatch):
    """penalties without an instantiated kernel are compiled at run time (csrc/wfa_rtc.cpp).  If that fails for a shape
    (WFA_HIP_RTC_FAIL=1 makes every compile fail after the availability probe has passed), the run is planned again without the
    run-time path: single calls, small and large batches, short and long reads all still equal the oracle."""
    import common
    from oracle import loader
    from pywfa_amd import datagen
    monkeypatch.setenv("WFA_HIP_RTC_FAIL", "1")
    for kw, (n, L, e) in [(dict(mismatch=7, gap_opening=5, gap_extension=3), (3000, 150, 0.03)),
                          (dict(mismatch=7, gap_opening=5, gap_extension=3, heuristic="adaptive", span="end-to-end"), (70000, 100, 0.02)),
                          (dict(mismatch=7, gap_opening=5, gap_extension=3, heuristic="adaptive", span="end-to-end"), (40, 1500, 0.05)),
                          (dict(mismatch=7, gap_opening=5, gap_extension=3), (1, 150, 0.03))]:
        batch = datagen.generate(n, L, e, 4242 + n)
        oc, nc = common.configs_pair(**dict(kw, scope=scope))
        o = loader.run(loader.oracle(), oc, batch, want_cigar=(scope == "full"))
        for resident in (True, False):
            score, status, cigars = common.gpu_run(nc, batch, scope == "full", resident=resident)
            common.assert_same(o, score, status, cigars, batch, f"rtc failure {kw} n={n} resident={resident}")
