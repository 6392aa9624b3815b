import numpy as np
import pytest

from pywfa_amd import datagen


def test_numpy_and_native_streams_agree():
    if not datagen._synth_lib():
        pytest.skip("csrc/libwfa_synth.so not built")
    a = datagen.generate(3000, 150, 0.02, 1002, use_native=False)
    b = datagen.generate(3000, 150, 0.02, 1002, use_native=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    c = datagen.generate(40, 2000, 0.08, 1003, first=17, use_native=True)
    d = datagen.generate(60, 2000, 0.08, 1003, use_native=False)
    for i in range(40):
        assert datagen.pair_strings(c, i) == datagen.pair_strings(d, 17 + i)


def test_error_model():
    b = datagen.generate(2000, 150, 0.06, 5)
    assert set(np.unique(b["seqs"][: 2000 * 150])) <= set(b"ACGT")
    assert abs(b["t_len"].mean() - 150) < 1.0          # insertions and deletions balance
    assert b["t_len"].std() > 1.0
    ident = datagen.generate(50, 150, 0.0, 5)
    for i in range(50):
        p, t = datagen.pair_strings(ident, i)
        assert p == t


def test_from_strings_uppercases():
    b = datagen.from_strings(["acgt", "AC"], ["ACGT", "ag"])
    assert datagen.pair_strings(b, 0) == ("ACGT", "ACGT")
    assert datagen.pair_strings(b, 1) == ("AC", "AG")


def test_from_strings_survives_upper_changing_a_length():
    """str.upper() can lengthen a string that still encodes as ASCII ('ß' -> 'SS'): the joined fast path must not
    shift the offsets of later pairs (the reference upper-cases every string on its own, align.pyx:432,435)."""
    from pywfa_amd import datagen
    b = datagen.from_strings(["ACßGT", "AAAA"], ["ACGT", "CCCC"])
    assert datagen.pair_strings(b, 0) == ("ACSSGT", "ACGT")
    assert datagen.pair_strings(b, 1) == ("AAAA", "CCCC")
    b = datagen.from_strings("ACßGT", ["ACGT", "CCßC", "GG"])
    assert [datagen.pair_strings(b, i) for i in range(3)] == [("ACSSGT", "ACGT"), ("ACSSGT", "CCSSC"), ("ACSSGT", "GG")]
    t = datagen.trim_text(datagen.generate(4, 300, 0.05, 9), 50)
    assert (t["t_len"] == datagen.generate(4, 300, 0.05, 9)["t_len"] - 100).all()
