"""One pair (empty pattern, 147-base text) under a run-time penalty shape through the single-call path (development aid)."""
import sys, os, faulthandler, subprocess
faulthandler.enable()
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1:
    import numpy as np, common
    from pywfa_amd import _native, datagen
    kw = eval(sys.argv[1])
    oc, nc = common.configs_pair(**kw)
    al = _native.Aligner(nc)
    t = "ACGTTGCAAGCTTAGGCATCGATCGGATTACAGGCATCGATTTACCGGATATCGGCTAGCTAGGATCCGATCGATTAGGCTTAACGGTATCGGATCGATTACGGCATTAGCCGATAGGCTAGCTAGGATCCGATCGATTAGGCTTAACGGTAGC"[:int(sys.argv[2])]
    r = al.align_pair(b"", t.encode(), kw.get("scope", "full") == "full")
    print("ok", kw, sys.argv[2], r[0], r[1], flush=True)
    al.close()
else:
    for env in ({}, {"WFA_HIP_NO_TINY_BAND": "1"}, {"WFA_HIP_NO_TINY": "1"}):
        for kw in ("dict(scope='full', mismatch=5)", "dict(scope='score', span='end-to-end', mismatch=5)", "dict(scope='full', mismatch=3, gap_opening=4, gap_extension=1)", "dict(scope='full')"):
            for L in ("147", "100", "60"):
                e = dict(os.environ, **env)
                out = subprocess.run([sys.executable, __file__, kw, L], env=e, capture_output=True, text=True)
                print(env, kw, L, "->", (out.stdout.strip() or "CRASH " + out.stderr.strip().splitlines()[0][:100]), flush=True)
