#!/usr/bin/env python3
"""Reproducer: with match < 0 and free begins the real WFA2-lib reads wavefront cells it never wrote, so a pair's score
depends on what the process aligned before (development aid; needs oracle/_ref, i.e. /root/reference).

    python tools/ref_endsfree_repro.py

A null step at a re-seeded score whose begin-free cell exists on one side only is allocated with lo = hi = +-j and
wf_elements_init_min = init_max = 0 (wavefront.c:107-108, wavefront_compute.c:214-254); when a later compute-next reads the
diagonals between 0 and j (wavefront_compute.c:490-520 believes them initialised) it gets whatever the slab held.  The pair
below scores 72 when it is the first alignment of the process and (on this machine) 73 after a batch of other pairs and the pair's own prefixes.  The
restatement in oracle/wfa_oracle.c reads NULL there and always gives 72."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import loader
from pywfa_amd import datagen

P = "GAAGGTTCTCACATTGCATACCATAGTTATGTGCATACACAAATTCTCTCGTTGAGGATGCCAGGAGTTACCATAGCATG"
T = "GACTGTTCTCACATTGCATACCATAGTTATGTGCATACACAAATTCTCTCGTTGAGGATGCCAGGAGTTACCATAGCATG"
cfg = loader.make_config(distance="affine", match=-1, mismatch=3, span="ends-free", pattern_begin_free=8, pattern_end_free=7,
                         text_begin_free=3, text_end_free=2, scope="score")
one = datagen.from_strings([P], [T])
first = int(loader.run(loader.reference(), cfg, one, want_cigar=False)["score"][0])
big = datagen.generate(1500, 150, 0.06, 77)
loader.run(loader.oracle(), cfg, big, want_cigar=False)       # (other heap traffic of the process)
loader.run(loader.reference(), cfg, big, want_cigar=False)
for L in range(10, 80, 5):   # (prefixes of the same pair: their wavefronts are what the slab hands out next)
    b = datagen.from_strings([P[:L]], [T[:L]])
    loader.run(loader.oracle(), cfg, b, want_cigar=False)
    loader.run(loader.reference(), cfg, b, want_cigar=False)
later = int(loader.run(loader.reference(), cfg, one, want_cigar=False)["score"][0])
oracle = int(loader.run(loader.oracle(), cfg, one, want_cigar=False)["score"][0])
print(f"reference, first alignment of the process: {first}; the same pair after 1500 other pairs and its own prefixes: {later}; oracle: {oracle}")
