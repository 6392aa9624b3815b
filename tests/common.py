"""Shared helpers of the parity tests (test infrastructure)."""
import json
import os

import numpy as np

from oracle import loader
from pywfa_amd import _native, datagen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def configs_pair(**kw):
    """(oracle Config, native Config) from pywfa-style kwargs."""
    oc = loader.make_config(**kw)
    nc = _native.Config()
    for name, _ in _native.Config._fields_:
        setattr(nc, name, getattr(oc, name))
    return oc, nc


def rle(b):
    out, i = [], 0
    b = bytes(b)
    while i < len(b):
        j = i
        while j < len(b) and b[j] == b[i]:
            j += 1
        out.append(f"{j - i}{chr(b[i])}")
        i = j
    return "".join(out)


def clamp_free(kw, batch):
    pl, tl = int(batch["p_len"].min()), int(batch["t_len"].min())
    kw = dict(kw)
    for k, lim in (("pattern_begin_free", pl), ("pattern_end_free", pl), ("text_begin_free", tl), ("text_end_free", tl)):
        if kw.get(k, 0) > lim:
            kw[k] = lim
    return kw


def gpu_run(nc, batch, full, resident):
    """Run the HIP path through the C ABI. Returns score, status, list of op-bytes or None."""
    al = _native.Aligner(nc)
    try:
        if resident:
            rb = al.batch(batch)
            rb.run()
            rb.sync()
            score, status, cig = rb.results(full)
            rb.close()
        else:
            score, status, cig = al.align_batch(batch, full)
    finally:
        al.close()
    cigars = None
    if full:
        ops, cbeg, clen = cig
        cigars = [ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() for i in range(len(score))]
    return score, status, cigars


def assert_same(o, score, status, cigars, batch, ctx):
    bad = np.flatnonzero((np.asarray(o["score"]) != score) | (np.asarray(o["status"]) != status))
    if bad.size:
        i = int(bad[0])
        p, t = datagen.pair_strings(batch, i)
        raise AssertionError(f"{ctx}: {bad.size} score/status mismatches; first pair {i}: expected "
                             f"({o['score'][i]}, {o['status'][i]}) got ({score[i]}, {status[i]})\nP={p[:200]}\nT={t[:200]}")
    if cigars is not None and o.get("cigars") is not None:
        for i, (a, b) in enumerate(zip(o["cigars"], cigars)):
            if a != b:
                p, t = datagen.pair_strings(batch, i)
                raise AssertionError(f"{ctx}: CIGAR mismatch at pair {i}: expected {rle(a)} got {rle(b)}\nP={p[:200]}\nT={t[:200]}")
