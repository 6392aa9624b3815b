"""bench.py's final stdout line: one strict-JSON object the driver can keep whole (round 3's 30 KB line was not
parsed).  These tests build records through the very formatter bench.py prints with (compact_record + format_line):
a realistic fat record (round 3's own, profiles/r03_bench.json), one with non-finite numbers, one far too long."""
import json
import os

import numpy as np
import pytest

import bench

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def strict_loads(line):
    def refuse(name):
        raise ValueError(f"non-JSON constant {name}")
    return json.loads(line, parse_constant=refuse)


def fat_record():
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:
        return json.load(f)


def check_line(line):
    assert "\n" not in line
    assert len(line.encode()) < 8000
    rec = strict_loads(line)
    for k in CONTRACT:
        assert k in rec, k
    assert isinstance(rec["config"]["workload"], str) and len(rec["config"]["workload"]) <= 120
    assert "model" not in rec["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    for v in list(rec["config"].values()) + list(rec["roofline"].values()):
        assert not isinstance(v, (dict, list)), "scalars only in config / roofline"
    return rec


def test_round3_fat_record_becomes_a_short_strict_line():
    full = fat_record()
    assert len(json.dumps(full)) > 20000          # the record that broke the driver's parser
    full["errors"] = []
    full["detail_file"] = "gpurun_out/bench_detail.json"
    rec = check_line(bench.format_line(bench.compact_record(full)))
    assert len(bench.format_line(bench.compact_record(full)).encode()) <= bench.LINE_LIMIT
    assert rec["value"] == pytest.approx(full["value"], rel=1e-5)
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert rec["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert rec["roofline"]["secondary_frac"] == pytest.approx(full["roofline"]["secondary_frac"], rel=1e-5)
    cb = rec["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["parity_mismatches"] == 0
    # the other half of the metric survives as scalars of config
    for k in ("c3_alignments_per_s", "c3_hbm_frac", "c3_parity_mismatches", "c4_exact_alignments_per_s",
              "end_to_end_alignments_per_s", "end_to_end_pcie_frac"):
        assert k in rec["config"], k
    assert rec["end_to_end"]["value"] == pytest.approx(full["end_to_end"]["value"], rel=1e-5)
    assert rec["detail"] == "gpurun_out/bench_detail.json" and rec["errors"] == []


def test_non_finite_numbers_never_reach_the_line():
    full = fat_record()
    full["roofline"]["frac"] = float("nan")
    full["roofline"]["achieved"] = float("inf")
    full["config"]["c3_hbm_frac"] = np.float64("nan")
    full["config"]["some_count"] = np.int64(7)
    full["vs_baseline"] = None
    rec = check_line(bench.format_line(bench.compact_record(full)))
    assert rec["roofline"]["frac"] is None and rec["roofline"]["achieved"] is None
    assert rec["config"]["c3_hbm_frac"] is None and rec["config"]["some_count"] == 7
    with pytest.raises(ValueError):                 # the formatter itself is strict: a raw NaN is an error, not output
        bench.format_line({"value": float("nan")})


def test_an_oversized_config_is_cut_not_the_contract():
    full = fat_record()
    for i in range(400):
        full["config"][f"configuration_number_{i:03d}_alignments_per_s"] = 1.0e6 + i
    full["errors"] = ["x" * 1000] * 20
    line = bench.format_line(bench.compact_record(full))
    rec = check_line(line)
    assert rec["config"]["truncated"] is True
    assert len(rec["errors"]) <= 8 and all(len(e) <= 200 for e in rec["errors"])
    assert "cpu_baseline" in rec and "roofline" in rec


def test_minimal_record_of_a_multi_gpu_run():
    """N > 1: no cpu_baseline, no extra configurations — the C3 leg's scalars ride in config."""
    full = {"metric": "pairwise alignments/sec", "value": 2.4e10, "unit": "alignments/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 3.3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "C2: 10000000 x 150bp pairs/GPU", "pairs_per_gpu": 10_000_000, "read_length": 150,
                       "parallelism": "pairs sharded over 8 GPU(s), no collective", "c3_alignments_per_s": 1.0e7,
                       "c3_hbm_frac": 0.018, "c3_invalid_transcripts": 0, "c3_pairs_per_gpu": 100000},
            "roofline": {"bound": "hbm", "achieved": 262.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0328, "traffic": None},
            "errors": []}
    rec = check_line(bench.format_line(bench.compact_record(full)))
    assert rec["n_gpus"] == 8 and rec["config"]["c3_alignments_per_s"] == 1.0e7 and "cpu_baseline" not in rec


def test_transcript_checker_accepts_the_oracle_and_rejects_damage():
    """transcripts_valid (the N > 1 legs' CPU-reference-free check) on op strings of the oracle."""
    from oracle import loader
    from pywfa_amd import datagen
    kw = dict(distance="affine", span="end-to-end", scope="full", heuristic="adaptive")
    batch = datagen.generate(40, 600, 0.08, 77)
    o = loader.run(loader.oracle(), loader.make_config(**kw), batch)
    n = len(o["score"])
    clen = np.array([len(c) for c in o["cigars"]], np.int32)
    cbeg = np.concatenate(([0], np.cumsum(clen)[:-1])).astype(np.int64)
    ops = np.frombuffer(b"".join(o["cigars"]), np.uint8).copy()
    score, status = np.asarray(o["score"]).copy(), np.asarray(o["status"]).copy()
    assert bench.transcripts_valid(batch, score, status, (ops, cbeg, clen), kw, sample=n) == 0
    score[3] -= 2                                   # penalty no longer equals -score
    ops[cbeg[5] + 10] = ord("X") if ops[cbeg[5] + 10] == ord("M") else ord("M")   # M over unequal / X over equal bases
    clen[7] -= 1                                    # does not consume the whole pattern / text
    assert bench.transcripts_valid(batch, score, status, (ops, cbeg, clen), kw, sample=n) == 3
