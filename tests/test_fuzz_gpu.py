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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "14", str(seed)], capture_output=True, text=True, timeout=1500)
    tail = "\n".join(out.stdout.splitlines()[-20:])
    assert out.returncode == 0 and "TOTAL BAD 0" in out.stdout, tail + out.stderr[-2000:]
