"""Replay a golden "python surface" case against a WavefrontAligner class.

Used twice: by tools/make_golden.py with the REFERENCE class (pywfa.align.WavefrontAligner, imported
from /root/reference in the build container) to record expected outputs, and by
tests/test_python_surface_gpu.py with pywfa_amd.WavefrontAligner to compare.  A case is plain data:

  {"name": str, "ctor": {kwargs}, "steps": [{"op": "align"|"call"|"set"|"get", ...}, ...]}
"""
import contextlib
import io
import os
import tempfile

RESULT_FIELDS = ("pattern_length", "text_length", "pattern_start", "pattern_end", "text_start",
                 "text_end", "cigartuples", "score", "pattern", "text", "status")


def _plain(v):
    if isinstance(v, tuple):
        return [_plain(x) for x in v]
    if isinstance(v, list):
        return [_plain(x) for x in v]
    return v


def snapshot_aligner(a):
    return {"status": a.status, "score": a.score, "cigarstring": a.cigarstring,
            "cigartuples": _plain(a.cigartuples), "locations": _plain(a.locations),
            "pattern_len": a.pattern_len, "text_len": a.text_len}


def snapshot_result(res):
    out = {f: _plain(getattr(res, f)) for f in RESULT_FIELDS}
    out["cigarstring"] = res.cigarstring
    out["aligned_pattern"] = res.aligned_pattern
    out["aligned_text"] = res.aligned_text
    out["repr"] = repr(res)
    out["str"] = str(res)
    if res.pattern and res.text:
        try:
            out["pretty"] = res.pretty
        except Exception as e:  # recorded as part of the behaviour
            out["pretty"] = f"EXC:{type(e).__name__}"
    return out


def _pretty_print(a):
    # cigar_print_pretty writes through C stdio in the reference: capture via a file
    fd, path = tempfile.mkstemp(suffix=".txt")
    os.close(fd)
    try:
        a.cigar_print_pretty(path)
        with open(path) as f:
            return f.read()
    finally:
        os.unlink(path)


def run_case(cls, case):
    """Execute the case's steps; return the list of per-step outputs."""
    outs = []
    try:
        a = cls(**case["ctor"])
    except Exception as e:
        return [{"ctor_exc": type(e).__name__}]
    for st in case["steps"]:
        op = st["op"]
        try:
            if op == "align":
                ret = a.wavefront_align(st["text"], st.get("pattern")) if st.get("pattern") is not None \
                    else a.wavefront_align(st["text"])
                outs.append({"ret": ret, "aligner": snapshot_aligner(a)})
            elif op == "call":
                res = a(st["text"], st.get("pattern"), **st.get("kwargs", {}))
                outs.append({"result": snapshot_result(res), "aligner": snapshot_aligner(a)})
            elif op == "set":
                setattr(a, st["name"], st["value"])
                outs.append({"ok": True})
            elif op == "get":
                outs.append({"value": _plain(getattr(a, st["name"]))})
            elif op == "pretty_print":
                outs.append({"text": _pretty_print(a)})
            else:
                raise KeyError(op)
        except Exception as e:
            outs.append({"exc": type(e).__name__})
    return outs
