"""The compiled host (pywfa_amd/host/_host.pyx + host_core.c): its batch marshalling equals
``datagen.from_strings`` byte for byte (the reference's ``upper().encode("ascii")`` per sequence, align.pyx:432,435), declines what
only the Python path may raise on, and — on the GPU box — serves ``wavefront_align_batch`` / ``wavefront_align`` with the same
results as the ctypes host."""
import random

import numpy as np
import pytest

from pywfa_amd import datagen
from pywfa_amd.host import build_host


@pytest.fixture(scope="module")
def host():
    build_host.build()
    from pywfa_amd.host import _host
    return _host


def same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), k


def test_marshalling_equals_the_python_path(host):
    rng = random.Random(5)

    def rs(n, low=False):
        s = "".join(rng.choice("ACGTN") for _ in range(n))
        return s.lower() if low else s
    for trial in range(120):
        n = rng.randrange(0, 40)
        pats = [rs(rng.randrange(0, 60), rng.random() < 0.3) for _ in range(n)]
        txts = [rs(rng.randrange(0, 60), rng.random() < 0.3) for _ in range(n)]
        if trial % 3 == 0 and n:
            txts[0] = txts[0].encode()          # bytes are taken as they are (no upper-casing), as in datagen.from_strings
        same(datagen.from_strings(pats, txts), host.from_strings(pats, txts))
        shared = "acGTn" + rs(10)
        same(datagen.from_strings(shared, txts), host.from_strings(shared, txts))
        same(datagen.from_strings(shared.encode(), txts), host.from_strings(shared.encode(), txts))
    big = [rs(150) for _ in range(20000)]       # (past the size where the OpenMP threads start)
    same(datagen.from_strings(big[:10000], big[10000:]), host.from_strings(big[:10000], big[10000:]))
    # buffers kept between calls (what WavefrontAligner does): a large batch, a small one, a larger one again
    scratch = {}
    for lo, hi in ((0, 6000), (100, 130), (0, 10000), (5, 6)):
        same(datagen.from_strings(big[lo:hi], big[10000 + lo:10000 + hi]), host.from_strings(big[lo:hi], big[10000 + lo:10000 + hi], scratch))
        same(datagen.from_strings(big[0], big[lo:hi]), host.from_strings(big[0], big[lo:hi], scratch))


def test_objects_only_the_python_path_may_judge(host):
    assert host.from_strings(["ACß"], ["AC"]) is None          # non-ASCII: encode("ascii") raises in the Python path
    assert host.from_strings("AC", [5]) is None                     # not a sequence: AttributeError there
    with pytest.raises(UnicodeEncodeError):
        datagen.from_strings(["ACé"], ["AC"])
    with pytest.raises(ValueError):
        host.from_strings(["A"], ["A", "C"])


@pytest.mark.gpu
def test_compiled_and_ctypes_hosts_agree_on_the_gpu(gpu, host, monkeypatch):
    import pywfa_amd
    from pywfa_amd import _native
    b = datagen.generate(3000, 150, 0.03, 99)
    pats = [datagen.pair_strings(b, i)[0] for i in range(3000)]
    txts = [datagen.pair_strings(b, i)[1].lower() if i % 5 == 0 else datagen.pair_strings(b, i)[1] for i in range(3000)]
    out = {}
    for which in ("compiled", "ctypes"):
        monkeypatch.setattr(_native, "_HOST", False if which == "compiled" else None)
        assert (_native.compiled_host() is not None) == (which == "compiled")
        a = pywfa_amd.WavefrontAligner(pats[0])
        r = a.wavefront_align_batch(txts, pats)
        singles = [(a.wavefront_align(txts[i], pats[i]), a.cigarstring, a.status) for i in range(0, 3000, 500)]
        rs_ = a.wavefront_align_batch(txts[:50])            # the cached pattern against every text
        out[which] = (r["score"].tolist(), r["status"].tolist(), list(r["cigarstrings"]), singles, rs_["score"].tolist())
    assert out["compiled"] == out["ctypes"]
    with pytest.raises(UnicodeEncodeError):
        pywfa_amd.WavefrontAligner("ACGT").wavefront_align_batch(["ACéT"])
