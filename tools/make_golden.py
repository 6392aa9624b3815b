#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference):
  * compiles pywfa's Cython host (pywfa/align.pyx) + the vendored WFA2-lib sources with cython and
    gcc directly (no reference build system, nothing copied into the repo) into a temp directory,
    imports it and records the outputs of the reference's own known-answer cases
    (/root/reference/pywfa/tests/test.py, README.rst) at the Python surface;
  * uses oracle/_ref (WFA2-lib C API) to record (status, score, CIGAR) for small seeded corpora.

The fixtures are DATA (inputs + expected outputs, including the sequences of the four FASTA files the
reference's tests hold); no reference source text is stored.

    python tools/make_golden.py
"""
import glob
import importlib
import json
import os
import subprocess
import sys
import sysconfig
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import loader  # noqa: E402
from pywfa_amd import datagen  # noqa: E402
import golden_runner  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference_extension():
    """cython + gcc on the reference sources where they lie -> temp dir containing package pywfa."""
    tmp = tempfile.mkdtemp(prefix="pywfa_ref_")
    pkg = os.path.join(tmp, "pywfa")
    os.makedirs(pkg)
    open(os.path.join(pkg, "__init__.py"), "w").close()
    W = os.path.join(REF, "pywfa", "WFA2_lib")
    c_out = os.path.join(pkg, "align.c")
    subprocess.check_call(["cython", "-3", os.path.join(REF, "pywfa", "align.pyx"), "-I", REF, "-o", c_out])
    srcs = []
    for d in ("wavefront", "alignment", "system", "utils"):
        srcs += sorted(glob.glob(os.path.join(W, d, "*.c")))
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    inc = ["-I" + REF, "-I" + W, "-I" + os.path.join(REF, "pywfa"), "-I" + sysconfig.get_paths()["include"]]
    for d in ("utils", "wavefront", "system", "alignment"):
        inc.append("-I" + os.path.join(W, d))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-w"] + inc + [c_out] + srcs +
                          ["-lm", "-o", os.path.join(pkg, "align" + ext)])
    sys.path.insert(0, tmp)
    mod = importlib.import_module("pywfa.align")
    return mod


def read_fasta(path):
    recs, name, seq = [], None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(seq)))
                name, seq = line[1:].split()[0], []
            elif line:
                seq.append(line)
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


def python_surface_cases():
    P1, T1 = "TCTTTACTCGCGCGTTGGAGAAATACAATAGT", "TCTATACTGCGCGTTTGGAGAAATAAAATAGT"
    P2, T2 = "AATTAATTTAAGTCTAGGCTACTTTCGGTACTTTGTTCTT", "AATTTAAGTCTAGGCTACTTTCGGTACTTTCTT"
    cases = []
    # test.py:16-51 test_affine (+ README.rst:34-42)
    cases.append({"name": "test_affine_1", "ctor": {"pattern": P1}, "expect": {"score": -24, "cigarstring": "3M1X4M1D7M1I9M1X6M", "status": 0},
                  "steps": [{"op": "align", "text": T1}, {"op": "pretty_print"}, {"op": "call", "text": T1}]})
    cases.append({"name": "test_affine_2_no_pattern_ctor", "ctor": {}, "expect": {"score": -24, "cigarstring": "3M1X4M1D7M1I9M1X6M", "status": 0},
                  "steps": [{"op": "call", "text": T1, "pattern": P1, "kwargs": {"clip_cigar": False}}]})
    cases.append({"name": "test_affine_3", "ctor": {}, "steps": [
        {"op": "call", "text": "TCTCCCCATACTGCGCGTTTGGAGAAATAAAA", "pattern": "TCTATACTGCGCGTTTGGAGAAATAAAA", "kwargs": {"clip_cigar": False}}]})
    # test.py:54-83 test_scope / test_supress_seqs
    cases.append({"name": "test_scope", "ctor": {"pattern": P1, "scope": "score"}, "expect": {"score": -24, "cigarstring": "", "status": 0},
                  "steps": [{"op": "call", "text": T1}]})
    cases.append({"name": "test_supress_seqs_score", "ctor": {"pattern": P1, "scope": "score"}, "expect": {"score": -24, "cigarstring": "", "status": 0},
                  "steps": [{"op": "call", "text": T1, "kwargs": {"supress_sequences": True}}]})
    cases.append({"name": "test_supress_seqs_full", "ctor": {"pattern": P1, "scope": "full"}, "expect": {"score": -24, "cigarstring": "3M1X4M1D7M1I9M1X6M", "status": 0},
                  "steps": [{"op": "call", "text": T1, "kwargs": {"supress_sequences": True}}]})
    # test.py:94-113
    kw = {"mismatch": 4, "gap_opening": 6, "gap_extension": 2}
    cases.append({"name": "test_end_to_end", "ctor": dict(pattern=P2, span="end-to-end", **kw), "expect": {"score": -26, "cigarstring": "4M4D26M3D3M"},
                  "steps": [{"op": "call", "text": T2}]})
    cases.append({"name": "test_ends_free", "ctor": dict(pattern=P2, span="ends-free", **kw), "expect": {"score": -26, "cigarstring": "4M4D26M3D3M"},
                  "steps": [{"op": "call", "text": T2, "kwargs": {"clip_cigar": True, "elide_mismatches": True, "min_aligned_bases_left": 5, "min_aligned_bases_right": 5}}]})
    # test.py:115-178 test_ends_free2
    ef2 = [("AAAAACCTTTTTAAAAAA", "GGCCAAAAACCAAAAAA"), ("AAAAACCTTTTTAAAAAA", "GGCCAAAAACCGGGGGGG"),
           ("AAAAACCGGGG", "AAAAACC"), ("AAAAACC", "AAAAACCGGGG"), ("GGGGAAAAACC", "AAAAACCGGGG"),
           ("AAAAACCGGGG", "GGGGAAAAACC"), ("GGGGAAAAACC", "AAAAACC"), ("GGGGAAAAACC", "CCCCCAAAAACC"),
           ("GGGGAAAAACCGGGGG", "CCCCCAAAAACCTTTTT"), ("AAAAACC", "CCCCCAAAAACCTTTTT")]
    for i, (p, t) in enumerate(ef2):
        cases.append({"name": f"test_ends_free2_{i}", "ctor": dict(pattern=p, span="ends-free", **kw),
                      "steps": [{"op": "call", "text": t}]})
    # test.py:180-194 test_heuristic
    for h in ("X-drop", "adaptive"):
        cases.append({"name": f"test_heuristic_{h}", "ctor": dict(pattern="AAAAACCAAAAAA", distance="affine", heuristic=h, **kw),
                      "steps": [{"op": "call", "text": "GGCCAAAAACCAAAAAA"}]})
    # README.rst:199-243
    cases.append({"name": "readme_clip", "ctor": {"pattern": "AAAAACCTTTTTAAAAAA"}, "steps": [
        {"op": "call", "text": "GGCCAAAAACCAAAAAA", "kwargs": {"clip_cigar": False}},
        {"op": "call", "text": "GGCCAAAAACCAAAAAA", "kwargs": {"clip_cigar": True}}]})
    P3, T3 = "AAAAAAAAAAAACCTTTTAAAAAAGAAAAAAA", "ACCCCCCCCCCCAAAAACCAAAAAAAAAAAAA"
    cases.append({"name": "readme_trim", "ctor": {"pattern": P3}, "steps": [
        {"op": "call", "text": T3, "kwargs": {"clip_cigar": False}},
        {"op": "call", "text": T3, "kwargs": {"clip_cigar": True, "min_aligned_bases_left": 5, "min_aligned_bases_right": 5}},
        {"op": "call", "text": T3, "kwargs": {"clip_cigar": True, "min_aligned_bases_left": 5, "min_aligned_bases_right": 5, "elide_mismatches": True}}]})
    # property surface, lower-case input, cached pattern re-use, score scope with clip/elide active (Q1)
    cases.append({"name": "props_and_reuse", "ctor": {"pattern": "acgtacgtaggt", "distance": "affine2p", "heuristic": "adaptive", "max_steps": 50}, "steps": [
        {"op": "get", "name": "distance"}, {"op": "get", "name": "heuristic"}, {"op": "get", "name": "scope"},
        {"op": "get", "name": "span"}, {"op": "get", "name": "memory_mode"}, {"op": "get", "name": "max_steps"},
        {"op": "get", "name": "mismatch_penalty"}, {"op": "get", "name": "gap_opening_penalty"},
        {"op": "get", "name": "gap_extension_penalty"}, {"op": "get", "name": "gap_opening2_penalty"},
        {"op": "get", "name": "gap_extension2_penalty"}, {"op": "get", "name": "match_score"},
        {"op": "get", "name": "min_wavefront_length"}, {"op": "get", "name": "max_distance_threshold"},
        {"op": "get", "name": "steps_between_cutoffs"}, {"op": "get", "name": "wildcard"},
        {"op": "get", "name": "pattern_begin_free"}, {"op": "get", "name": "text_end_free"},
        {"op": "align", "text": "acgtaggtaggt"}, {"op": "align", "text": "ACGTACGTAGGT"},
        {"op": "call", "text": "ttacgtacgtaggtaa"}, {"op": "align", "text": "GGGG", "pattern": "CCCCCCCC"},
        {"op": "call", "text": "GGGG"}]})
    cases.append({"name": "score_scope_clip_elide", "ctor": {"pattern": P3, "scope": "score"}, "steps": [
        {"op": "call", "text": T3, "kwargs": {"clip_cigar": True, "elide_mismatches": True}}]})
    cases.append({"name": "max_steps_hit", "ctor": {"pattern": P2, "max_steps": 5}, "steps": [{"op": "call", "text": T2}]})
    cases.append({"name": "xdrop_dropped", "ctor": {"pattern": P3, "heuristic": "X-drop", "xdrop": 5}, "steps": [{"op": "call", "text": T3}]})
    cases.append({"name": "wildcard", "ctor": {"pattern": "ACGTNNGTACGT", "wildcard": "N"}, "steps": [
        {"op": "call", "text": "ACGTCCGTACGT"}, {"op": "call", "text": "ACGTCCGNACGT"}]})
    cases.append({"name": "endsfree_sizes", "ctor": dict(pattern="GGGGAAAAACCGGGGG", pattern_begin_free=4, pattern_end_free=5, text_begin_free=5, text_end_free=5), "steps": [
        {"op": "call", "text": "CCCCCAAAAACCTTTTT"}, {"op": "get", "name": "pattern_begin_free"}]})
    for d in ("indel", "levenshtein", "linear"):
        cases.append({"name": f"distance_{d}", "ctor": {"pattern": P1, "distance": d}, "steps": [
            {"op": "get", "name": "distance"}, {"op": "call", "text": T1}, {"op": "align", "text": T2, "pattern": P2}]})
    cases.append({"name": "empty_text", "ctor": {"pattern": "ACGT"}, "steps": [{"op": "call", "text": ""}]})
    cases.append({"name": "pattern_none", "ctor": {}, "steps": [{"op": "call", "text": "ACGT"}]})
    cases.append({"name": "bad_scope", "ctor": {"scope": "half"}, "steps": []})
    cases.append({"name": "bad_span", "ctor": {"span": "local"}, "steps": []})
    cases.append({"name": "bad_heuristic", "ctor": {"heuristic": "zdrop"}, "steps": []})
    cases.append({"name": "bad_distance", "ctor": {"distance": "hamming"}, "steps": []})
    cases.append({"name": "bad_memory_mode", "ctor": {"memory_mode": "tiny"}, "steps": []})
    cases.append({"name": "bad_wildcard", "ctor": {"wildcard": "NN"}, "steps": []})
    return cases


def fasta_cases():
    T = os.path.join(REF, "pywfa", "tests")
    reads = read_fasta(os.path.join(T, "short.fa"))
    refs = read_fasta(os.path.join(T, "short.reference.fa"))
    cases = []
    for (rn, rs), (fn, fs) in zip(reads, refs):
        text, pattern = rs.upper(), fs.upper()
        cases.append({"name": f"test_short:{rn}", "ctor": {"mismatch": 5, "gap_opening": 6, "gap_extension": 2},
                      "steps": [{"op": "call", "text": text, "pattern": pattern}]})
        cases.append({"name": f"test_short2p:{rn}", "ctor": {"distance": "affine2p", "mismatch": 5, "gap_opening": 6, "gap_extension": 2},
                      "steps": [{"op": "call", "text": text, "pattern": pattern, "kwargs": {"clip_cigar": True, "elide_mismatches": True}}]})
    lr = read_fasta(os.path.join(T, "long.fa"))
    lf = read_fasta(os.path.join(T, "long.reference.fa"))
    for (rn, rs), (fn, fs) in zip(lr, lf):
        text, pattern = rs.upper(), fs.upper()
        lt, lp = int(len(text) / 2), int(len(pattern) / 2)
        cases.append({"name": f"test_long:{rn}", "ctor": {"distance": "affine", "mismatch": 4, "gap_opening": 6, "gap_extension": 2,
                                                         "pattern_begin_free": lp, "pattern_end_free": lp, "text_begin_free": lt, "text_end_free": lt},
                      "steps": [{"op": "call", "text": text, "pattern": pattern, "kwargs": {"clip_cigar": True}}]})
    return cases


def helper_vectors(mod):
    """I/O vectors of the module-level helpers (align.pyx:183-295) and AlignmentResult."""
    rng = np.random.default_rng(11)
    vec = {"cigartuples_to_str": [], "elide": [], "clip": []}
    for _ in range(60):
        n = int(rng.integers(0, 9))
        ct = [(int(rng.choice([0, 1, 2, 8, 4])), int(rng.integers(1, 12))) for _ in range(n)]
        vec["cigartuples_to_str"].append({"in": ct, "out": mod.cigartuples_to_str(ct)})
        vec["elide"].append({"in": ct, "out": [list(x) for x in mod.elide_mismatches_from_cigar(ct)]})
    for _ in range(120):
        n = int(rng.integers(1, 9))
        ct = [(int(rng.choice([0, 1, 2, 8])), int(rng.integers(1, 12))) for _ in range(n)]
        pl = sum(l for o, l in ct if o in (0, 2, 8))
        tl = sum(l for o, l in ct if o in (0, 1, 8))
        ts0 = int(rng.integers(0, 3))
        left, right = int(rng.integers(1, 8)), int(rng.integers(1, 8))
        res = mod.AlignmentResult(pl, tl, 0, pl, ts0, tl, list(ct), -7, "P" * pl, "T" * tl, 0)
        out = mod.clip_cigartuples(res, left, right)
        vec["clip"].append({"ct": ct, "pl": pl, "tl": tl, "ts0": ts0, "left": left, "right": right,
                            "out": {"cigartuples": [list(x) for x in out.cigartuples], "text_start": out.text_start,
                                    "text_end": out.text_end, "pattern_start": out.pattern_start, "pattern_end": out.pattern_end}})
    return vec


def rle(b):
    out, i = [], 0
    while i < len(b):
        j = i
        while j < len(b) and b[j] == b[i]:
            j += 1
        out.append(f"{j - i}{chr(b[i])}")
        i = j
    return "".join(out)


def c_level_vectors():
    """(status, score, CIGAR) of oracle/_ref on small seeded corpora x configurations."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import validate_oracle as vo
    corpora = {"special": vo.corpus_special()}
    for L, e, n in ((150, 0.02, 64), (150, 0.15, 48), (1000, 0.08, 10)):
        corpora[f"L{L}_e{e}"] = datagen.generate(n, L, e, 4242 + L)
    cfgs = vo.configs(True)[::3] + [dict(distance="indel"), dict(distance="levenshtein", span="end-to-end"),
                                    dict(distance="levenshtein", heuristic="adaptive", scope="score"),
                                    dict(distance="linear", mismatch=3, gap_extension=5), dict(distance="linear", match=-1, span="end-to-end"),
                                    dict(distance="affine", span="end-to-end", scope="score"),
                                    dict(distance="affine2p", span="ends-free", scope="full", pattern_begin_free=8, pattern_end_free=7, text_begin_free=3, text_end_free=2)]
    out = {"corpora": {}, "runs": []}
    for name, b in corpora.items():
        n = len(b["p_len"])
        pairs = [list(datagen.pair_strings(b, i)) for i in range(n)]
        if name == "special":
            pairs = pairs[:160] + pairs[300:360] + pairs[620:700]
        out["corpora"][name] = pairs
    for name, pairs in out["corpora"].items():
        b = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
        for kw in cfgs:
            kw = vo.clamp_free(kw, b)
            r = vo.run_reference(kw, b)
            if r is None:
                continue
            out["runs"].append({"corpus": name, "config": kw, "score": [int(x) for x in r["score"]],
                                "status": [int(x) for x in r["status"]],
                                "cigar": [rle(c) for c in r["cigars"]] if r["cigars"] is not None else None})
    return out


def biwfa_vectors():
    """(status, score, CIGAR) of oracle/_ref in its ultralow (BiWFA) mode, scope=full: reads that never split (<= 100 bases: the
    score stays unset), reads that split once, and reads that split several levels deep."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import validate_oracle as vo
    corpora = {"special": vo.corpus_special()}
    for L, e, n in ((60, 0.1, 40), (150, 0.02, 48), (150, 0.2, 40), (1500, 0.1, 8), (5000, 0.12, 3)):
        corpora[f"L{L}_e{e}"] = datagen.generate(n, L, e, 5151 + L)
    cfgs = [dict(span="end-to-end"), dict(), dict(distance="affine2p"), dict(distance="levenshtein", span="end-to-end"), dict(distance="indel"),
            dict(distance="linear", mismatch=3, gap_extension=5), dict(match=-1, span="end-to-end"), dict(wildcard="N")]
    out = {"corpora": {}, "runs": []}
    for name, b in corpora.items():
        n = len(b["p_len"])
        pairs = [list(datagen.pair_strings(b, i)) for i in range(n)]
        if name == "special":
            pairs = pairs[:120] + pairs[300:340] + pairs[620:680]
        out["corpora"][name] = pairs
    for name, pairs in out["corpora"].items():
        b = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
        for kw in cfgs:
            kw = dict(kw, scope="full", memory_mode="biwfa")
            r = vo.run_reference(kw, b)
            if r is None:
                continue
            out["runs"].append({"corpus": name, "config": kw, "score": [int(x) for x in r["score"]],
                                "status": [int(x) for x in r["status"]], "cigar": [rle(c) for c in r["cigars"]]})
    # round 3: a step limit (counted over the forward + reverse scores, wavefront_bialign.c:475,513; the base cases carry it too,
    # wavefront_bialigner.c:168-174), both scopes; reads of <= 100 bases under large penalties (the top-level base case has no
    # score bound)
    limited = [dict(span="end-to-end"), dict(distance="affine2p"), dict(distance="levenshtein", span="end-to-end"), dict(match=-1, span="end-to-end")]
    for name in ("special", "L60_e0.1", "L150_e0.2", "L1500_e0.1"):
        pairs = out["corpora"][name]
        b = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
        for kw in limited:
            for ms in (20, 150, 700):
                for scope in ("full", "score"):
                    kw2 = dict(kw, scope=scope, memory_mode="biwfa", max_steps=ms)
                    r = vo.run_reference(kw2, b)
                    if r is None:
                        continue
                    out["runs"].append({"corpus": name, "config": kw2, "score": [int(x) for x in r["score"]], "status": [int(x) for x in r["status"]],
                                        "cigar": [rle(c) for c in r["cigars"]] if r["cigars"] is not None else None})
    # round 4: a heuristic, inherited by the forward / reverse aligner of every breakpoint search (wavefront_bialigner.c:53,161-166), both scopes
    heur = [dict(heuristic="adaptive"), dict(heuristic="adaptive", min_wavefront_length=5, max_distance_threshold=15, steps_between_cutoffs=3),
            dict(heuristic="X-drop", xdrop=400), dict(heuristic="X-drop", xdrop=100, match=-1), dict(heuristic="adaptive", distance="affine2p"),
            dict(heuristic="adaptive", distance="levenshtein")]
    for name in ("special", "L150_e0.2", "L1500_e0.1", "L5000_e0.12"):
        pairs = out["corpora"][name]
        b = datagen.from_strings([p for p, _ in pairs], [t for _, t in pairs])
        for kw in heur:
            for scope in ("full", "score"):
                kw2 = dict(kw, scope=scope, memory_mode="biwfa", span="end-to-end")
                r = vo.run_reference(kw2, b)
                if r is None:
                    continue
                out["runs"].append({"corpus": name, "config": kw2, "score": [int(x) for x in r["score"]], "status": [int(x) for x in r["status"]],
                                    "cigar": [rle(c) for c in r["cigars"]] if r["cigars"] is not None else None})
    rng = np.random.default_rng(77)
    unrelated = [["".join(rng.choice(list("ACGT"), size=int(rng.integers(60, 101)))), "".join(rng.choice(list("ACGT"), size=int(rng.integers(60, 101))))]
                 for _ in range(24)]
    out["corpora"]["unrelated_le100"] = unrelated
    b = datagen.from_strings([p for p, _ in unrelated], [t for _, t in unrelated])
    for kw in (dict(match=-1, span="end-to-end"), dict(mismatch=6, gap_opening=2, gap_extension=3), dict(distance="affine2p", match=-2), dict(match=-1, max_steps=600)):
        kw2 = dict(kw, scope="full", memory_mode="biwfa")
        r = vo.run_reference(kw2, b)
        if r is not None:
            out["runs"].append({"corpus": "unrelated_le100", "config": kw2, "score": [int(x) for x in r["score"]], "status": [int(x) for x in r["status"]],
                                "cigar": [rle(c) for c in r["cigars"]]})
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    loader.build()
    if len(sys.argv) > 1 and sys.argv[1] == "biwfa":   # only the BiWFA vectors (round 2)
        os.makedirs(GOLD, exist_ok=True)
        with open(os.path.join(GOLD, "biwfa.json"), "w") as f:
            json.dump(biwfa_vectors(), f, indent=0)
        print("biwfa.json", os.path.getsize(os.path.join(GOLD, "biwfa.json")))
        return
    mod = build_reference_extension()
    os.makedirs(GOLD, exist_ok=True)
    surface = []
    for case in python_surface_cases() + fasta_cases():
        outs = golden_runner.run_case(mod.WavefrontAligner, case)
        surface.append({"case": case, "expected": outs})
        # cross-check the values the reference's own tests assert
        exp = case.get("expect")
        if exp:
            last = [o for o in outs if "aligner" in o][-1]["aligner"]
            for k, v in exp.items():
                assert last[k] == v, (case["name"], k, last[k], v)
    with open(os.path.join(GOLD, "python_surface.json"), "w") as f:
        json.dump(surface, f, indent=0)
    with open(os.path.join(GOLD, "helpers.json"), "w") as f:
        json.dump(helper_vectors(mod), f, indent=0)
    with open(os.path.join(GOLD, "c_level.json"), "w") as f:
        json.dump(c_level_vectors(), f, indent=0)
    with open(os.path.join(GOLD, "biwfa.json"), "w") as f:
        json.dump(biwfa_vectors(), f, indent=0)
    for fn in sorted(os.listdir(GOLD)):
        print(fn, os.path.getsize(os.path.join(GOLD, fn)))


if __name__ == "__main__":
    main()
