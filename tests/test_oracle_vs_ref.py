"""Where oracle/_ref (the real WFA2-lib, compiled from /root/reference) is present: the restatement
must agree with it bit-for-bit on fresh seeded corpora.  (tools/validate_oracle.py is the long form.)"""
import numpy as np
import pytest

import common
from oracle import loader
from pywfa_amd import datagen

pytestmark = pytest.mark.skipif(not loader.have_reference(), reason="oracle/_ref not built (needs /root/reference)")

CONFIGS = [
    dict(span="end-to-end", scope="score"),
    dict(span="end-to-end", scope="full"),
    dict(scope="full"),
    dict(distance="affine2p", scope="full"),
    dict(distance="affine2p", span="ends-free", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2),
    dict(heuristic="adaptive"),
    dict(heuristic="X-drop", xdrop=100),
    dict(heuristic="X-drop", xdrop=20, scope="score"),
    dict(max_steps=10),
    dict(match=-1, span="end-to-end"),
    dict(distance="indel"), dict(distance="levenshtein", heuristic="adaptive"), dict(distance="linear", mismatch=3, gap_extension=5),
    dict(distance="linear", match=-1, span="end-to-end", scope="score"),
]


@pytest.mark.parametrize("cfg_idx", range(len(CONFIGS)))
@pytest.mark.parametrize("shape", [(3000, 150, 0.02), (600, 150, 0.15), (60, 1000, 0.08)])
def test_oracle_equals_reference(cfg_idx, shape):
    n, L, e = shape
    batch = datagen.generate(n, L, e, 900 + cfg_idx)
    kw = common.clamp_free(CONFIGS[cfg_idx], batch)
    cfg = loader.make_config(**kw)
    r = loader.run(loader.reference(), cfg, batch)
    o = loader.run(loader.oracle(), cfg, batch)
    common.assert_same(r, o["score"], o["status"], o["cigars"], batch, f"oracle vs reference {kw}")
