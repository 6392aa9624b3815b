#!/usr/bin/env python3
"""bench.py — pairwise alignments/s of the batched wavefront-alignment hot path on MI355X.

Headline workload (BASELINE.json configs[1], "C2"): 10 M x 150 bp synthetic short-read pairs at 2 % error
(seed 1002), gap-affine 0/4/6/2, end-to-end, scope=score.  One "step" = one pass of the alignment kernels
over the whole batch, with the 2-bit packed sequences already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P]

N > 1: one rank per GPU (the driver launches torch.distributed.run; `python bench.py --gpus N` without a
WORLD_SIZE spawns it itself); every rank aligns its own P pairs (weak scaling, pairs are independent: no
collective on the data path); barrier + device sync on both sides of the timed region, MAX over ranks, rank 0
prints ONE JSON line.

The LAST stdout line is one compact strict-JSON record (about 5 KB; tests/test_bench_line.py): the contract keys, `config` (workload + one
scalar triple per extra configuration), `roofline`, `cpu_baseline`, `end_to_end` as scalars.  The fat record described below goes to
bench_detail.json (and gpurun_out/bench_detail.json); a failed configuration, a parity mismatch or an invalid transcript ends the run non-zero.
N > 1: every rank also times a C3 leg (10 kb, wf-adaptive, full CIGAR; `config.c3_alignments_per_s`).  --multi: only the C5 leg through ONE
process's wfa_hip_multi_align_batch over every visible device.

Besides the contract keys the detail record carries (rank 0, N = 1):
  roofline      HBM roofline of the C2 step (HIP events on the launch stream), + "secondary": the on-chip bound
                from the committed counter passes (profiles/), + "traffic" with its provenance
  end_to_end    the PCIe-inclusive rate of the same batch: host ASCII in -> host results out (wfa_hip_align_batch)
  offsets_per_s wavefront offsets (all components) the reference algorithm computes for these pairs, per second
  cpu_baseline  the real WFA2-lib (oracle/_ref) on the same pairs, 1 thread (+ all host threads), and a PARITY CHECK of
                the scores it computed against the GPU's: parity_checked_pairs / parity_mismatches
  extra.configs the other BASELINE configurations (C1 150 bp full CIGAR, C3 10 kb adaptive full CIGAR, C4 10 kb affine2p
                ends-free full CIGAR) on stated prefixes, each with kernel_ms, alignments/s, roofline, offsets/s, a
                1-thread reference baseline and a parity check against it
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PCIE_PEAK_GBS = 63.0    # same guide: PCIe Gen5 x16


def host_cpu():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def kernel_source_hash():
    """Hash of the kernel sources: ties counter files under profiles/ to the build they were collected on."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "pywfa_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def cpu_reference(batch, cfg_kw, n_max, budget_s, want_cigar):
    """The CPU reference on a bounded prefix (1 thread).  Returns (info dict, n, result dict)."""
    from oracle import loader
    from pywfa_amd import datagen
    if loader.have_reference():
        fn, kind = loader.reference(), "reference"
    else:
        fn, kind = loader.oracle(), "port"
    cfg = loader.make_config(**cfg_kw)
    n_all = len(batch["p_len"])
    probe = int(min(n_all, n_max, max(4, n_max // 100)))
    t0 = time.perf_counter()
    res = loader.run(fn, cfg, datagen.subset(batch, np.arange(probe)), want_cigar=want_cigar)
    dt = max(time.perf_counter() - t0, 1e-6)
    n = int(min(n_all, n_max, max(probe, budget_s * probe / dt)))
    if n > probe:
        t0 = time.perf_counter()
        res = loader.run(fn, cfg, datagen.subset(batch, np.arange(n)), want_cigar=want_cigar)
        dt = time.perf_counter() - t0
    info = {"value": n / dt, "unit": "alignments/s", "cores": 1, "kind": kind,
            "sample": f"first {n} pairs of the same batch, 1 thread, {dt:.1f} s; host CPU: {host_cpu()} ({os.cpu_count()} logical cores)",
            "library": os.path.basename(loader.reference_path() or "liboracle.so")}
    return info, n, res


def parity(res, n, score, status, cig):
    """Mismatches between the CPU reference's results on the first n pairs and the GPU's."""
    bad = (np.asarray(res["score"])[:n] != score[:n]) | (np.asarray(res["status"])[:n] != status[:n])
    if cig is not None and res.get("cigars") is not None:
        ops, cbeg, clen = cig
        for i in range(n):
            if not bad[i] and ops[cbeg[i]:cbeg[i] + clen[i]].tobytes() != res["cigars"][i]:
                bad[i] = True
    return int(bad.sum())


def work_counts(batch, cfg_kw, n):
    """Offsets the reference algorithm computes per pair (oracle counters on a small sample)."""
    from oracle import loader
    from pywfa_amd import datagen
    n = int(min(n, len(batch["p_len"])))
    loader.oracle_counters(True)
    loader.run(loader.oracle(), loader.make_config(**dict(cfg_kw, scope="score")), datagen.subset(batch, np.arange(n)), want_cigar=False)
    m, allc, bases = loader.oracle_counters()
    return {"m_offsets_per_pair": m / n, "offsets_per_pair": allc / n, "bases_compared_per_pair": bases / n, "sample_pairs": n}


def run_resident(al, batch, steps, warmup, want_cigar, barrier=None, min_seconds=0.0):
    """Warm-up, then `steps` timed passes over the resident batch.  Returns timings and results.
    min_seconds (extra configurations only): short steps are repeated until the timed region lasts about that long (<= 50 steps),
    so that one slow enqueue on a busy host does not decide a 3-step measurement."""
    t0 = time.perf_counter()
    rb = al.batch(batch)
    t_upload = time.perf_counter() - t0
    for _ in range(warmup):
        rb.run()
    rb.sync()
    if min_seconds > 0.0:
        t0 = time.perf_counter()
        rb.run(); rb.sync()
        est = max(time.perf_counter() - t0, 1e-5)
        steps = max(steps, min(50, int(min_seconds / est) + 1))
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        rb.run()
    rb.sync()  # device sync of the stream the kernels run on
    if barrier:
        barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, kernel_pairs = rb.last_kernel()  # mean HIP-event time per step over the timed steps
    score, status, cig = rb.results(want_cigar)
    out = {"elapsed": elapsed, "kernel_ms": kernel_ms, "upload_s": t_upload, "score": score, "status": status, "cig": cig,
           "fallback": rb.fallback_pairs(), "io_bytes": rb.algorithmic_bytes(), "steps": steps}
    rb.close()
    return out


def counter_file(name, src_hash):
    """profiles/traffic_<name>.json (tools/make_profiles.py): HBM bytes per run and the dominant kernel's on-chip counters,
    used only when collected on this kernel source.  Returns (traffic bytes per run or None, secondary or None, provenance)."""
    path = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    try:
        with open(path) as f:
            tj = json.load(f)
    except (OSError, ValueError):
        return None, None, None
    same = tj.get("kernel_source_hash") == src_hash
    prov = {"file": f"profiles/traffic_{name}.json", "collected_on_kernel_source": tj.get("kernel_source_hash"),
            "same_kernel_source": same, "pairs_profiled": tj.get("pairs"), "counter_files": tj.get("source")}
    if not same:
        return None, None, prov
    return tj.get("hbm_bytes_per_pair"), tj.get("secondary"), prov


def extra_config(name, n, length, error, seed, cfg_kw, scheme, survey_bytes, trim=0, cpu_pairs=400, cpu_budget=4.0, env=None,
                 mt_parity_pairs=0, steps=3):
    """One of the other BASELINE configurations on a stated prefix: kernel time, roofline, CPU baseline, parity.
    env: library knobs for this configuration (read when its aligner is created); mt_parity_pairs: check that many pairs
    (scores, statuses, op strings) against the reference run on all host threads, beyond the 1-thread timing sample."""
    from pywfa_amd import _native, datagen
    from oracle import loader
    batch = datagen.generate(n, length, error, seed)
    if trim:
        batch = datagen.trim_text(batch, trim)
    oc = loader.make_config(**cfg_kw)
    nc = _native.Config()
    for fname, _ in _native.Config._fields_:
        setattr(nc, fname, getattr(oc, fname))
    full = oc.scope == 1
    for k_, v_ in (env or {}).items():
        os.environ[k_] = v_
    try:
        al = _native.Aligner(nc, 0)
    finally:
        for k_ in (env or {}):
            del os.environ[k_]
    r = run_resident(al, batch, steps, 1, full, min_seconds=0.1)
    steps = r["steps"]
    al.close()
    wc = work_counts(batch, cfg_kw, 8 if length >= 5000 else 2000)
    # algorithmic HBM bytes per pair: SURVEY.md §8(d)'s per-unit figure for this configuration and history scheme
    # (packed sequences in + results / op bytes out + the wavefront history written once)
    bytes_pair = float(survey_bytes)
    achieved = bytes_pair * n / (r["kernel_ms"] * 1e-3) / 1e9
    cpu, n_cpu, res = cpu_reference(batch, cfg_kw, cpu_pairs, cpu_budget, full)
    n_bad = parity(res, n_cpu, r["score"], r["status"], r["cig"])
    if mt_parity_pairs > n_cpu and loader.have_reference():
        n_mt = int(min(n, mt_parity_pairs))
        res_mt = loader.reference_mt_full(oc, datagen.subset(batch, np.arange(n_mt)), os.cpu_count() or 1, want_cigar=full)
        n_bad = parity(res_mt, n_mt, r["score"], r["status"], r["cig"])
        n_cpu = n_mt
    rate = n / (r["elapsed"] / steps)
    traffic_pair, secondary, prov = counter_file(name, kernel_source_hash())
    return {"name": name, "pairs": n, "read_length": length, "error": error, "seed": seed, "config": cfg_kw,
            "history_scheme": scheme if (full and length > 1000) else None,
            "kernel_ms": r["kernel_ms"], "ms_per_step": r["elapsed"] / steps * 1e3, "steps": steps, "alignments_per_s": rate,
            "offsets_per_s": rate * wc["offsets_per_pair"], "work": wc,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "bytes_per_pair": bytes_pair, "bytes_per_pair_source": "SURVEY.md §8(d)",
                         "io_bytes_per_pair_measured": r["io_bytes"] / n,
                         "traffic": None if traffic_pair is None else traffic_pair * n, "traffic_bytes_per_pair": traffic_pair,
                         "traffic_provenance": prov, "secondary": secondary},
            "completed": int((r["status"] == 0).sum()), "handed_to_general_kernel": int(r["fallback"]),
            "cpu_baseline": cpu, "speedup_vs_1_thread": rate / cpu["value"],
            "parity_checked_pairs": n_cpu, "parity_mismatches": n_bad}


LINE_LIMIT = 6000   # bytes of the final stdout line (the driver keeps an 8 KB tail; round 3's 30 KB line was not parsed)
WORKLOAD_LIMIT = 120


def _num(v):
    """Finite, short numbers: 6 significant digits are enough for a rate, and NaN / Infinity are not JSON."""
    if isinstance(v, (bool, type(None), str)):
        return v
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.6g}")
    return v


def _scalars(d, keys=None):
    """The scalar entries of a dict (optionally only `keys`), numbers shortened; nested objects are dropped."""
    out = {}
    for k, v in d.items():
        if keys is not None and k not in keys:
            continue
        if isinstance(v, (dict, list, tuple)):
            continue
        out[k] = _num(v)
    return out


def compact_record(full):
    """The record of the final stdout line: the contract keys, `config` (workload <= 120 chars + one scalar triple per
    extra configuration), `roofline`, `cpu_baseline`, `end_to_end` — scalars only.  Everything else (per-configuration
    rooflines, counters' provenance, prose) stays in the detail file."""
    rec = {k: _num(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                          "scaling", "vs_baseline", "dtype", "data")}
    cfg = _scalars(full.get("config", {}))
    if isinstance(cfg.get("workload"), str):
        cfg["workload"] = cfg["workload"][:WORKLOAD_LIMIT]
    rec["config"] = cfg
    roof = full.get("roofline") or {}
    rec["roofline"] = _scalars(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "secondary_bound", "secondary_frac",
                                      "algorithmic_bytes_per_launch", "bytes_per_pair", "kernel_ms", "kernel"))
    rec["roofline"].setdefault("traffic", None)
    if "cpu_baseline" in full:
        cb = _scalars(full["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "library", "parity_checked_pairs",
                                             "parity_mismatches", "all_threads_value", "all_threads_threads"))
        if isinstance(cb.get("sample"), str):
            cb["sample"] = cb["sample"][:160]
        rec["cpu_baseline"] = cb
    if "end_to_end" in full:
        rec["end_to_end"] = _scalars(full["end_to_end"], ("value", "unit", "statistic", "seconds_per_batch", "seconds_min", "seconds_max",
                                                          "value_best_call", "pcie_gb_s", "pcie_frac", "ascii_gb_s_consumed", "includes_pilot"))
        host = full["end_to_end"].get("host") or {}
        rec["end_to_end"]["gpu_numa_node"] = host.get("gpu_numa_node")
        rec["end_to_end"]["workers_bound"] = host.get("workers_bound")
        bound = full["end_to_end"].get("workers_bound_to_gpu_numa_node") or {}
        rec["end_to_end"]["value_workers_bound_to_gpu_node"] = _num(bound.get("value")) if bound else None
        two = full["end_to_end"].get("packed2bits_input") or {}
        rec["end_to_end"]["value_2bit_input"] = _num(two.get("value")) if two else None
    if "offsets_per_s" in full:
        rec["offsets_per_s"] = _num(full["offsets_per_s"])
    rec["errors"] = [str(e)[:200] for e in full.get("errors", [])][:8]
    rec["detail"] = full.get("detail_file")
    rec["kernel_source_hash"] = (full.get("extra") or {}).get("kernel_source_hash")
    return rec


def format_line(rec, limit=LINE_LIMIT):
    """One strict-JSON line of at most `limit` bytes.  If the record is too long the per-configuration scalars of `config`
    go first (longest names first) — the contract keys, `roofline` and `cpu_baseline` always stay."""
    def dump(r):
        return json.dumps(r, allow_nan=False, separators=(", ", ": "))
    line = dump(rec)
    if len(line.encode()) > limit:
        rec = json.loads(line)
        core = ("workload", "pairs_per_gpu", "read_length", "parallelism")
        extras = sorted((k for k in rec["config"] if k not in core), key=lambda k: (-len(k), k))
        rec["config"]["truncated"] = True
        for k in extras:
            del rec["config"][k]
            line = dump(rec)
            if len(line.encode()) <= limit:
                break
    if len(line.encode()) > limit:
        raise ValueError(f"bench line is {len(line.encode())} bytes, limit {limit}")
    return line


def write_detail(full):
    """The fat record (every configuration's roofline, counters' provenance, CPU samples) goes to a side file, and to
    gpurun_out/ when that exists so that it travels back from the GPU box; never to stdout."""
    name = "bench_detail.json"
    text = json.dumps(full, default=lambda o: _num(o) if isinstance(o, (np.integer, np.floating)) else str(o))
    written = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            if d.endswith("gpurun_out"):
                os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as f:
                f.write(text + "\n")
            written = written or os.path.relpath(os.path.join(d, name), ROOT)
        except OSError:
            pass
    return written


def native_config(**kw):
    """pywfa's constructor kwargs -> wfa_hip_config_t (defaults = wfa_hip_config_default = align.pyx:309-334)."""
    from pywfa_amd import _native
    c = _native.default_config()
    for k, v in kw.items():
        if k == "distance":
            c.distance = _native.DIST[v]
        elif k == "scope":
            c.scope = _native.SCOPE[v]
        elif k == "span":
            c.span = _native.SPAN[v]
        elif k == "heuristic":
            c.heuristic = _native.HEUR[v]
        elif k == "memory_mode":
            c.memory_mode = _native.MEM[v]
        elif k == "wildcard":
            c.wildcard = -1 if v is None else ord(v.upper())
        else:
            setattr(c, k, int(v))
    return c


def transcripts_valid(batch, score, status, cig, cfg_kw, sample=2000):
    """Size-independent check of a full-CIGAR run without a CPU reference (N > 1 legs): every completed pair's op string
    consumes exactly plen / tlen bases (inside the free ends), M only over equal bases, and — match = 0 — its gap-affine
    penalty equals -score.  Checked on an evenly spaced sample; returns the number of violations."""
    ops, cbeg, clen = cig
    n = len(score)
    x, o, e = cfg_kw.get("mismatch", 4), cfg_kw.get("gap_opening", 6), cfg_kw.get("gap_extension", 2)
    bad = 0
    for i in np.linspace(0, n - 1, min(sample, n)).astype(np.int64):
        if status[i] != 0:
            bad += 1
            continue
        s = ops[cbeg[i]:cbeg[i] + clen[i]]
        nm, nx, ni, nd = (int((s == c).sum()) for c in (77, 88, 73, 68))
        if nm + nx + nd != int(batch["p_len"][i]) or nm + nx + ni != int(batch["t_len"][i]):
            bad += 1
            continue
        pi = np.cumsum(s != 73) - 1 + int(batch["p_off"][i])   # pattern base under each op (I consumes none)
        ti = np.cumsum(s != 68) - 1 + int(batch["t_off"][i])
        m, xm = s == 77, s == 88
        if (batch["seqs"][pi[m]] != batch["seqs"][ti[m]]).any() or (batch["seqs"][pi[xm]] == batch["seqs"][ti[xm]]).any():
            bad += 1
            continue
        if cfg_kw.get("distance", "affine") == "affine" and cfg_kw.get("span", "ends-free") == "end-to-end":
            g = (s == 73) | (s == 68)
            opens = int((g & ~np.concatenate(([False], s[:-1] == s[1:]))).sum())
            if nx * x + opens * o + (ni + nd) * e != -int(score[i]):
                bad += 1
    return bad


def shard_first(rank, pairs_per_gpu):
    """Weak scaling: rank r owns pairs [r*P, (r+1)*P) of the one seeded stream (no overlap, no gaps)."""
    return rank * pairs_per_gpu


def dist_setup(backend, local_rank):
    """One process per GPU; RCCL ("nccl") on the GPU box, gloo in the CPU tests."""
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def dist_barrier(dist, backend):
    if dist is None:
        return
    if backend == "nccl":
        import torch
        torch.cuda.synchronize()
    dist.barrier()


def dist_max(dist, backend, value):
    """MAX over ranks of a host float (the step time)."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dist_all(dist, backend, value):
    """Every rank's host float, in rank order (per-rank rates: min / max / imbalance of a weak-scaling run)."""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def rank_spread(rates):
    """min / max / imbalance (max over min, minus one) of per-rank rates."""
    lo, hi = min(rates), max(rates)
    return {"per_rank": rates, "min": lo, "max": hi, "imbalance": (hi / lo - 1.0) if lo > 0 else None}


def dist_sum(dist, backend, value):
    """SUM over ranks of a host number (pairs completed, violations)."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


C3_KW = dict(distance="affine", span="end-to-end", scope="full", heuristic="adaptive")
C5_XDROP_KW = dict(distance="affine", span="end-to-end", scope="full", heuristic="X-drop", xdrop=20)   # C5 as BASELINE writes it
C5_ADAPT_KW = dict(distance="affine", span="end-to-end", scope="full", heuristic="adaptive")


def c3_leg(rank, world, local_rank, pairs, steps, dist, backend):
    """The 10 kb half of the metric at any N: every rank aligns its own `pairs` of the C3 stream (10 kb, 8 %, wf-adaptive,
    full CIGAR on the device), barrier on both sides, MAX time over ranks, pairs summed.  No CPU reference here: the op strings
    are checked through their invariants (transcripts_valid)."""
    from pywfa_amd import _native, datagen
    batch = datagen.generate(pairs, 10000, 0.08, datagen.SEEDS["C3"], first=shard_first(rank, pairs))
    al = _native.Aligner(native_config(**C3_KW), device=local_rank)
    r = run_resident(al, batch, steps, 1, True, barrier=lambda: dist_barrier(dist, backend))
    al.close()
    elapsed = dist_max(dist, backend, r["elapsed"])
    spread = rank_spread([pairs * steps / t for t in dist_all(dist, backend, r["elapsed"])])
    bad = transcripts_valid(batch, r["score"], r["status"], r["cig"], C3_KW, sample=500)
    bad = dist_sum(dist, backend, bad)
    done = dist_sum(dist, backend, int((r["status"] == 0).sum()))
    rate = pairs * world * steps / elapsed
    bytes_pair = 114e3   # SURVEY §8(d), piggy-back history
    return {"pairs_per_gpu": pairs, "steps": steps, "ms_per_step": elapsed / steps * 1e3, "alignments_per_s": rate,
            "kernel_ms_rank0": r["kernel_ms"], "hbm_frac": bytes_pair * pairs / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "completed": int(done), "invalid_transcripts": int(bad), "checked_per_rank": min(500, pairs),
            "rank_alignments_per_s_min": spread["min"], "rank_alignments_per_s_max": spread["max"], "rank_imbalance": spread["imbalance"],
            "per_rank_alignments_per_s": spread["per_rank"]}


def python_host_leg(n=1_000_000):
    """The drop-in surface itself: `WavefrontAligner.wavefront_align_batch(texts, patterns)` on `n` Python str pairs
    of 150 bp (scope = score; best of three calls after a warm-up: the first large call pins the upload ring), and the per-call cost of
    `wavefront_align(text)` — pywfa's own usage pattern.  Scores are checked against the resident C2 path's."""
    import pywfa_amd
    from pywfa_amd import _native, datagen
    b = datagen.generate(n, 150, 0.02, datagen.SEEDS["C2"])
    blob = b["seqs"].tobytes().decode("ascii")
    po, pl, to, tl = (b[k].tolist() for k in ("p_off", "p_len", "t_off", "t_len"))
    pats = [blob[po[i]:po[i] + pl[i]] for i in range(n)]
    txts = [blob[to[i]:to[i] + tl[i]] for i in range(n)]
    a = pywfa_amd.WavefrontAligner(scope="score", span="end-to-end")
    a.wavefront_align_batch(txts[:1000], pats[:1000])
    a.wavefront_align_batch(txts, pats)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        r = a.wavefront_align_batch(txts, pats)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    al = _native.Aligner(a._cfg, 0)
    ref, _, _ = al.align_batch(b, False)
    al.close()
    m = 2000
    t0 = time.perf_counter()
    for i in range(m):
        a.wavefront_align(txts[i], pats[i])
    single = (time.perf_counter() - t0) / m
    return {"pairs": n, "seconds_per_call": best, "pairs_per_s": n / best, "single_call_us": single * 1e6,
            "compiled_host": _native.compiled_host() is not None, "score_mismatches": int((r["score"] != ref).sum()),
            "what": "WavefrontAligner.wavefront_align_batch(list[str], list[str]), scope=score, 150 bp; wavefront_align(text, pattern) per call"}


def multi_leg(pairs_per_device, length=100000, calls=2):
    """C5 through ONE process's wfa_hip_multi_align_batch over every visible device (the product's own sharding):
    host ASCII in -> host results out, so this rate includes PCIe.  X-drop(20) as BASELINE writes C5 (every pair is dropped
    after a few steps, SURVEY Q2: status 1, INT32_MIN, empty CIGAR) and wf-adaptive (completes)."""
    from pywfa_amd import _native, datagen
    ndev = _native.lib().wfa_hip_device_count()
    devices = list(range(ndev))
    pairs_per_device = max(64, min(pairs_per_device, 16384 // max(ndev, 1)))   # (200 KB of ASCII and of op bytes per pair on the host)
    n = pairs_per_device * ndev
    batch = datagen.generate(n, length, 0.08, datagen.SEEDS["C5"])
    out = {"devices": ndev, "pairs": n, "read_length": length}
    for name, kw in (("xdrop", C5_XDROP_KW), ("adaptive", C5_ADAPT_KW)):
        ma = _native.MultiAligner(native_config(**kw), devices)
        best = None
        score = status = cig = None
        for _ in range(calls):
            # (the previous call's results go first: giving 1.6 GB of op bytes back to the system is ~50 ms that belongs to no call)
            del score, status, cig
            t0 = time.perf_counter()
            score, status, cig = ma.align_batch(batch, True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ma.close()
        ops, cbeg, clen = cig
        if name == "xdrop":
            bad = int(((status != 1) | (score != -2**31) | (clen != 0)).sum())
        else:
            bad = transcripts_valid(batch, score, status, cig, kw, sample=64)
        out[name] = {"alignments_per_s": n / best, "seconds_per_call": best, "completed": int((status == 0).sum()),
                     "dropped": int((status == 1).sum()), "violations": bad}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=10_000_000, help="pairs per GPU")
    ap.add_argument("--length", type=int, default=150)
    ap.add_argument("--error", type=float, default=0.02)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--c3-pairs", type=int, default=100_000, help="N > 1: 10 kb pairs per GPU of the C3 leg")
    ap.add_argument("--c3-leg", action="store_true", help="run the C3 leg of the N > 1 runs at N = 1 too (it has no CPU reference: transcripts are validated)")
    ap.add_argument("--multi", action="store_true",
                    help="only the C5 leg: one process, wfa_hip_multi_align_batch over every visible device")
    ap.add_argument("--multi-pairs", type=int, default=8192, help="100 kb pairs per device of the C5 leg (at most 16 384 pairs in all; "
                    "BASELINE's C5 is 12 500 per device; 4 096 pairs leave the first stage's launches half empty: 23 k against 36 k aln/s)")
    args = ap.parse_args()

    if args.multi:
        m = multi_leg(args.multi_pairs)
        full = {"metric": "pairwise alignments/sec", "value": m["adaptive"]["alignments_per_s"], "unit": "alignments/s",
                "n_gpus": m["devices"], "steps": 1, "warmup": 1, "ms_per_step": m["adaptive"]["seconds_per_call"] * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "config": {"workload": f"C5: {m['pairs']} x 100kb pairs, 8% error, wf-adaptive, full CIGAR, one process, "
                                       f"wfa_hip_multi_align_batch over {m['devices']} device(s), host in -> host out",
                           "pairs_per_gpu": m["pairs"] // max(m["devices"], 1), "read_length": 100000,
                           "parallelism": f"contiguous shards over {m['devices']} device(s), no collective",
                           "c5_multi_adaptive_alignments_per_s": m["adaptive"]["alignments_per_s"],
                           "c5_multi_adaptive_violations": m["adaptive"]["violations"],
                           "c5_multi_xdrop_alignments_per_s": m["xdrop"]["alignments_per_s"],
                           "c5_multi_xdrop_violations": m["xdrop"]["violations"]},
                "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
                "extra": {"multi": m, "kernel_source_hash": kernel_source_hash()},
                "errors": [f"{k}: {m[k]['violations']} violations" for k in ("xdrop", "adaptive") if m[k]["violations"]]}
        full["detail_file"] = write_detail(full)
        print(format_line(compact_record(full)), flush=True)
        sys.exit(1 if full["errors"] else 0)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by hand: start one rank per GPU as a child (nothing has touched the GPU yet) and
        # leave with its return code
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dist = None
    # (development aid: WFA_BENCH_BACKEND=gloo WFA_BENCH_SHARE_DEVICE=1 runs the N > 1 path with every rank on device 0 of a 1-GPU box —
    # everything but RCCL itself; the driver's runs use RCCL and one device per rank)
    backend = os.environ.get("WFA_BENCH_BACKEND", "nccl")
    if os.environ.get("WFA_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        dist = dist_setup(backend, local_rank)
        # the job the driver asked for: N ranks, one per GPU, over RCCL (backend "nccl" on ROCm)
        assert dist.get_world_size() == args.gpus and dist.get_rank() == rank, (dist.get_world_size(), args.gpus, dist.get_rank(), rank)
        assert dist.get_backend() == backend
    n_gpus = world

    from pywfa_amd import _native, datagen

    cfg_kw = dict(distance="affine", match=0, mismatch=4, gap_opening=6, gap_extension=2,
                  span="end-to-end", scope="score")
    cfg = _native.default_config()
    cfg.span, cfg.scope = _native.SPAN["end-to-end"], _native.SCOPE["score"]

    # each rank owns the pairs [rank*P, (rank+1)*P) of the seed-1002 stream (weak scaling)
    t0 = time.perf_counter()
    batch = datagen.generate(args.pairs, args.length, args.error, datagen.SEEDS["C2"], first=shard_first(rank, args.pairs))
    t_gen = time.perf_counter() - t0

    al = _native.Aligner(cfg, device=local_rank)
    r = run_resident(al, batch, args.steps, args.warmup, False, barrier=lambda: dist_barrier(dist, backend))
    elapsed = dist_max(dist, backend, r["elapsed"])
    c2_spread = rank_spread([args.pairs * args.steps / t for t in dist_all(dist, backend, r["elapsed"])])
    kernel_ms, score, status = r["kernel_ms"], r["score"], r["status"]
    # PCIe-inclusive rate (host ASCII in -> host results out): eight calls, the first (it pins and sizes the staging) left out; the line
    # carries the MEDIAN as the figure, with the minimum and every call's time beside it (a best-of-five hid a 2x spread)
    # (results into caller-owned arrays, as a C caller has them: fresh 2 x 40 MB NumPy arrays per call cost ~10 ms of page faults)
    def time_e2e(a, calls, what=None):
        what = batch if what is None else what
        ts = []
        outs = (np.zeros(args.pairs, np.int32), np.zeros(args.pairs, np.int32))
        for _ in range(calls + 1):
            t0 = time.perf_counter()
            s2, st2, _ = a.align_batch(what, False, out=outs)
            ts.append(time.perf_counter() - t0)
        assert np.array_equal(s2, score) and np.array_equal(st2, status)
        ts = ts[1:]
        return {"median": float(np.median(ts)), "min": float(min(ts)), "max": float(max(ts)), "calls": [round(x, 5) for x in ts]}
    dist_barrier(dist, backend)
    e2e_stats = time_e2e(al, 7)
    upload_info = al.upload_info()
    t_e2e = dist_max(dist, backend, e2e_stats["median"])   # (N > 1: every rank's call at once, the host cores shared: the slowest rank counts)
    # a caller that holds 2-bit reads already (wfa_hip_align_batch_packed2bits): no host packing, a quarter of the bytes
    # (N > 1 too, every rank at once, the slowest counted: with ASCII input eight ranks read 8 x 300 B per pair from host memory — the
    # 2-bit entry and the resident rate are the figures that can scale with the devices)
    pk = datagen.to_packed2bits(batch)
    dist_barrier(dist, backend)
    e2e_2bit_stats = time_e2e(al, 5, pk)
    t_e2e_2bit = dist_max(dist, backend, e2e_2bit_stats["median"])
    del pk
    al.close()
    # the same call with the upload workers bound to the GPU's NUMA node (WFA_HIP_NUMA=1, round 5's behaviour): the A/B leg VERDICT r05 asked
    # for — the default leaves the scheduler alone (csrc/wfa_hip.hip: numa_lookup has the measurements)
    e2e_bound_stats = None
    if rank == 0 and n_gpus == 1 and args.pairs >= 262144 and upload_info["gpu_node_cpus"] > 0:
        os.environ["WFA_HIP_NUMA"] = "1"
        alb = _native.Aligner(cfg, device=local_rank)
        e2e_bound_stats = time_e2e(alb, 5)
        e2e_bound_stats["upload_info"] = alb.upload_info()
        alb.close()
        del os.environ["WFA_HIP_NUMA"]
    # the same call with the host packer off (ASCII over PCIe + device pack kernel), for the record
    t_e2e_ascii = None
    if rank == 0 and n_gpus == 1 and args.pairs >= 262144:
        os.environ["WFA_HIP_HOST_PACK"] = "0"
        al0 = _native.Aligner(cfg, device=local_rank)   # (the knobs are read when the aligner is created)
        del os.environ["WFA_HIP_HOST_PACK"]
        t_e2e_ascii = time_e2e(al0, 2)["median"]
        al0.close()

    c3 = None
    if args.c3_leg or (n_gpus > 1 and not args.no_extra_configs):
        c3 = c3_leg(rank, world, local_rank, args.c3_pairs, 3, dist, backend)

    if rank == 0:
        errors = []
        src_hash = kernel_source_hash()
        # HBM traffic of one launch and the on-chip counters, from the PMC passes committed under profiles/ (bench.py
        # cannot collect counters itself): used only when they were collected on this workload AND this kernel source
        traffic, provenance, secondary = None, None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic_c2.json")) as f:
                tj = json.load(f)
            w = tj["workload"]
            same_wl = (w["pairs_per_gpu"], w["read_length"], w["error"]) == (args.pairs, args.length, args.error)
            same_src = tj.get("kernel_source_hash") == src_hash
            provenance = {"file": "profiles/traffic_c2.json", "collected_on_kernel_source": tj.get("kernel_source_hash"),
                          "this_build_kernel_source": src_hash, "same_workload": same_wl, "same_kernel_source": same_src,
                          "git_head_when_collected": tj.get("git_head"), "counter_files": tj.get("source")}
            if same_wl and same_src:
                traffic = tj["hbm_bytes_per_launch"]
                secondary = tj.get("secondary")
        except (OSError, KeyError, ValueError):
            pass
        total_pairs = args.pairs * n_gpus * args.steps
        value = total_pairs / elapsed
        alg_bytes = r["io_bytes"]
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        wc = work_counts(batch, cfg_kw, 20000)
        e2e_rate = args.pairs / t_e2e
        ascii_bytes = float(batch["p_len"].sum() + batch["t_len"].sum()) / args.pairs + 8
        host_packed = args.pairs >= 262144   # (csrc/wfa_hip.hip batch_build: the large-batch form)
        words = float((((batch["p_len"].astype(np.int64) + 15) >> 4) + ((batch["t_len"].astype(np.int64) + 15) >> 4)).sum()) / args.pairs
        # host-packed: the 2-bit words + 4 B of 16-bit lengths per pair up (round 6: the 16 B metadata records are rebuilt on the device),
        # 8 B of results down; small batches: the ASCII bytes + 16 B metadata + 16 B byte offsets
        sent_bytes = (4 * words + 4 + 8) if host_packed else (ascii_bytes + 32)
        out = {
            "metric": "pairwise alignments/sec",
            "value": value,
            "unit": "alignments/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # (the C2 kernel keeps wavefront offsets as packed int16 pairs — reads <= 512 bp, range-guarded, every score
            # checked against the reference in this run; the reference's wf_offset_t is int32, wavefront_offset.h:38)
            "dtype": "int16x2 offsets (reads <= 512 bp), int32 results",
            "data": "synthetic",
            "config": {"workload": f"C2: {args.pairs} x {args.length}bp pairs/GPU, {args.error * 100:g}% error, gap-affine 0/4/6/2, "
                                   "end-to-end, scope=score, 2-bit reads resident in HBM",
                       # (the first stage's width is chosen by a pilot on 8192 pairs once per resident batch, outside the timed steps;
                       #  wfa_hip_align_batch pays it on every call: the end_to_end figures below include it)
                       "pilot_in_timed_region": False, "end_to_end_includes_pilot": True,
                       "pairs_per_gpu": args.pairs, "read_length": args.length, "parallelism": f"pairs sharded over {n_gpus} GPU(s), no collective",
                       # per-rank rates of the timed steps (weak scaling: every rank the same work; `value` is all pairs over the slowest rank's time)
                       "rank_alignments_per_s_min": c2_spread["min"], "rank_alignments_per_s_max": c2_spread["max"],
                       "rank_imbalance": c2_spread["imbalance"],
                       # `value` is the HBM-resident rate; the PCIe-inclusive rate of the same batch (host ASCII in -> host results out, every
                       # rank's call at once, the slowest rank counted) is never `value` but belongs beside it (full detail: "end_to_end")
                       "end_to_end_alignments_per_s": e2e_rate * n_gpus, "end_to_end_seconds_per_batch": t_e2e,
                       "end_to_end_pcie_frac": e2e_rate * sent_bytes / 1e9 / PCIE_PEAK_GBS,
                       "end_to_end_2bit_input_alignments_per_s": None if t_e2e_2bit is None else args.pairs * n_gpus / t_e2e_2bit},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_provenance": provenance,
                         "secondary": secondary,
                         "secondary_bound": None if not secondary else secondary.get("bound"),
                         "secondary_frac": None if not secondary else secondary.get("frac"),
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_pair": alg_bytes / max(args.pairs, 1),
                         "kernel_ms": kernel_ms, "kernel": "wfa alignment kernels of one step (HIP events on the launch stream)"},
            "offsets_per_s": value * wc["offsets_per_pair"],
            "end_to_end": {"value": e2e_rate * n_gpus, "unit": "alignments/s", "seconds_per_batch": t_e2e,
                           "what": "wfa_hip_align_batch: host ASCII in -> host scores/status out (host threads pack to 2 bits into the pinned "
                                   "upload ring, DMA, align, download); results into caller-owned arrays",
                           "statistic": "median of 7 calls after one warm-up call", "seconds_min": e2e_stats["min"], "seconds_max": e2e_stats["max"],
                           "seconds_calls": e2e_stats["calls"], "value_best_call": args.pairs / e2e_stats["min"] * n_gpus,
                           "host": upload_info,
                           "workers_bound_to_gpu_numa_node": None if e2e_bound_stats is None else {
                               "value": args.pairs / e2e_bound_stats["median"], "seconds_per_batch": e2e_bound_stats["median"],
                               "seconds_min": e2e_bound_stats["min"], "seconds_calls": e2e_bound_stats["calls"], "host": e2e_bound_stats["upload_info"],
                               "what": "WFA_HIP_NUMA=1: the spawned upload workers bound to the CPUs of the GPU's NUMA node (round 5's default)"},
                           "includes_pilot": True, "ascii_bytes_per_pair": ascii_bytes, "ascii_gb_s_consumed": e2e_rate * ascii_bytes / 1e9,
                           "pcie_bytes_per_pair": sent_bytes, "pcie_gb_s": e2e_rate * sent_bytes / 1e9,
                           "pcie_frac": e2e_rate * sent_bytes / 1e9 / PCIE_PEAK_GBS,
                           "packed2bits_input": None if t_e2e_2bit is None else {
                               "value": args.pairs * n_gpus / t_e2e_2bit, "seconds_per_batch": t_e2e_2bit, "seconds_min": e2e_2bit_stats["min"],
                               "seconds_calls": e2e_2bit_stats["calls"],
                               "what": "wfa_hip_align_batch_packed2bits: the caller holds 2-bit reads (4 bases per byte); the upload workers re-base them "
                                       "to whole words on their way into the pinned ring (round 6): the same bytes cross PCIe as for ASCII input",
                               "pcie_bytes_per_pair": sent_bytes},
                           "ascii_upload": None if t_e2e_ascii is None else {
                               "value": args.pairs / t_e2e_ascii, "seconds_per_batch": t_e2e_ascii,
                               "what": "WFA_HIP_HOST_PACK=0: the ASCII blob crosses PCIe, the device packs it",
                               "pcie_gb_s": args.pairs / t_e2e_ascii * (ascii_bytes + 32) / 1e9,
                               "pcie_frac": args.pairs / t_e2e_ascii * (ascii_bytes + 32) / 1e9 / PCIE_PEAK_GBS}},
            "extra": {"mean_score": float(score.mean()), "completed": int((status == 0).sum()),
                      "fallback_pairs": int(r["fallback"]), "datagen_s": t_gen, "upload_pack_s": r["upload_s"],
                      "work_per_pair": wc, "kernel_source_hash": src_hash},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import loader
            cpu, n_cpu, res = cpu_reference(batch, cfg_kw, args.pairs, 12.0, False)
            cpu["parity_checked_pairs"] = n_cpu
            cpu["parity_mismatches"] = parity(res, n_cpu, score, status, None)
            if cpu["kind"] == "reference":
                # context only: the same library on every host thread (one aligner object per thread); its scores are
                # checked too
                # (steady state: every thread walks its slice of the batch `repeat` times with one aligner, the repeat count
                # chosen so that the run lasts >= 5 s; nothing is subtracted: thread start, the aligners' set-up and the
                # first-touch faults are inside the time, they just no longer dominate it)
                try:
                    nt = os.cpu_count() or 1
                    ocfg = loader.make_config(**cfg_kw)
                    t0 = time.perf_counter()
                    loader.reference_mt(ocfg, batch, nt, 1)
                    t_1 = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    loader.reference_mt(ocfg, batch, nt, 4)
                    t_4 = time.perf_counter() - t0
                    per_pass = max((t_4 - t_1) / 3.0, 1e-4)
                    repeat = int(min(2000, max(4, np.ceil(6.0 / per_pass))))
                    t0 = time.perf_counter()
                    rmt = loader.reference_mt(ocfg, batch, nt, repeat)
                    dt_mt = time.perf_counter() - t0
                    cpu["all_threads"] = {"value": args.pairs * repeat / dt_mt, "threads": nt, "repeat": repeat, "seconds": dt_mt,
                                          "one_pass_seconds": t_1, "four_pass_seconds": t_4,
                                          "sample": f"all {args.pairs} pairs x {repeat} passes, one aligner per thread, {dt_mt:.1f} s (a single pass "
                                                    f"takes {t_1:.2f} s, set-up dominated: {args.pairs / t_1 / 1e6:.1f} M/s)",
                                          "parity_checked_pairs": args.pairs, "parity_mismatches": parity(rmt, args.pairs, score, status, None)}
                    cpu["all_threads_value"] = cpu["all_threads"]["value"]
                    cpu["all_threads_threads"] = nt
                except Exception as e:  # the single-thread figure above is the reported baseline
                    cpu["all_threads"] = {"error": str(e)}
            out["cpu_baseline"] = cpu
        if n_gpus == 1 and not args.no_extra_configs:
            C4 = dict(distance="affine2p", span="ends-free", pattern_begin_free=100, pattern_end_free=100, scope="full")
            xs = []
            plan = [
                dict(name="C1", n=1_000_000, length=150, error=0.02, seed=datagen.SEEDS["C1"], cfg_kw=dict(scope="full"), scheme="explicit",
                     survey_bytes=236, cpu_pairs=200000, cpu_budget=3.0),
                # (long reads keep the piggy-back history by default; WFA_HIP_BAND_PB=0 = the explicit offsets, for the record)
                dict(name="C3", n=100_000, length=10000, error=0.08, seed=datagen.SEEDS["C3"],
                     cfg_kw=dict(span="end-to-end", scope="full", heuristic="adaptive"), scheme="piggyback", survey_bytes=114e3),
                # (the same path on a batch that fills the chip: a 1 M-pair launch of the lane kernel is 3 900 waves)
                dict(name="C1-10M-pairs", n=10_000_000, length=150, error=0.02, seed=datagen.SEEDS["C1"], cfg_kw=dict(scope="full"), scheme="explicit",
                     survey_bytes=236, cpu_pairs=100000, cpu_budget=2.0),
                # (100 k pairs: the last stage of this cascade is one alignment's latency long whatever the batch size)
                dict(name="C4-adaptive", n=100_000, length=10000, error=0.08, seed=datagen.SEEDS["C4"], cfg_kw=dict(C4, heuristic="adaptive"),
                     scheme="piggyback", survey_bytes=114e3 * 5 / 3, trim=50, cpu_pairs=100),
                dict(name="C4-exact", n=4096, length=10000, error=0.08, seed=datagen.SEEDS["C4"], cfg_kw=C4, scheme="piggyback", survey_bytes=54e6,
                     trim=50, cpu_pairs=8, cpu_budget=3.0, mt_parity_pairs=128),
                dict(name="exact-10kb-score", n=8192, length=10000, error=0.08, seed=datagen.SEEDS["C3"], cfg_kw=dict(span="end-to-end", scope="score"),
                     scheme="none", survey_bytes=2 * 2500 + 8, cpu_pairs=40, cpu_budget=3.0, mt_parity_pairs=256),
                dict(name="exact-10kb-full", n=8192, length=10000, error=0.08, seed=datagen.SEEDS["C3"], cfg_kw=dict(span="end-to-end", scope="full"),
                     scheme="piggyback", survey_bytes=5.1e6 + 2 * 2500 + 10_800, cpu_pairs=20, cpu_budget=3.0, mt_parity_pairs=128),
                dict(name="C5-adaptive", n=8192, length=100000, error=0.08, seed=datagen.SEEDS["C5"],
                     cfg_kw=dict(span="end-to-end", scope="full", heuristic="adaptive"), scheme="piggyback", survey_bytes=1.14e6, cpu_pairs=4,
                     cpu_budget=2.0, mt_parity_pairs=128, steps=2),
                # wf-adaptive on short reads: the general form of the lane kernel goes first where its pilot finds that
                # few pairs outgrow its 16 slots (0.5 % divergence), the banded kernel takes the batch otherwise (2 %)
                dict(name="150bp-adaptive-2pct", n=2_000_000, length=150, error=0.02, seed=datagen.SEEDS["C2"],
                     cfg_kw=dict(span="end-to-end", scope="score", heuristic="adaptive"), scheme="none", survey_bytes=84, cpu_pairs=200000, cpu_budget=2.0),
                dict(name="150bp-adaptive-0.5pct", n=2_000_000, length=150, error=0.005, seed=datagen.SEEDS["C2"],
                     cfg_kw=dict(span="end-to-end", scope="score", heuristic="adaptive"), scheme="none", survey_bytes=84, cpu_pairs=200000, cpu_budget=2.0),
                # penalties the library has no instantiation of (round 4: the register kernels are compiled for them at run time, csrc/wfa_rtc.cpp;
                # mismatch=5 is the reference's own test's, pywfa/tests/test.py:229), and configurations mapped to gap-affine with a translated
                # score (match < 0: Eizenga-rescaled 10/12/5; levenshtein / gap-linear in score scope)
                dict(name="C2-mismatch5", n=2_000_000, length=150, error=0.02, seed=datagen.SEEDS["C2"],
                     cfg_kw=dict(span="end-to-end", scope="score", mismatch=5), scheme="none", survey_bytes=84, cpu_pairs=200000, cpu_budget=2.0),
                dict(name="C2-match-1", n=2_000_000, length=150, error=0.02, seed=datagen.SEEDS["C2"],
                     cfg_kw=dict(span="end-to-end", scope="score", match=-1), scheme="none", survey_bytes=84, cpu_pairs=200000, cpu_budget=2.0),
                dict(name="150bp-levenshtein", n=2_000_000, length=150, error=0.02, seed=datagen.SEEDS["C2"],
                     cfg_kw=dict(distance="levenshtein", span="end-to-end", scope="score"), scheme="none", survey_bytes=84, cpu_pairs=200000, cpu_budget=2.0),
                # (round 6: the one-component distances WITH CIGARs on the register kernels' LIN form; round 5: general kernel, 0.17 G aln/s)
                dict(name="150bp-levenshtein-full", n=2_000_000, length=150, error=0.02, seed=datagen.SEEDS["C2"],
                     cfg_kw=dict(distance="levenshtein", span="end-to-end", scope="full"), scheme="explicit", survey_bytes=236, cpu_pairs=100000, cpu_budget=2.0),
                dict(name="150bp-linear", n=2_000_000, length=150, error=0.02, seed=datagen.SEEDS["C2"],
                     cfg_kw=dict(distance="linear", span="end-to-end", scope="score"), scheme="none", survey_bytes=84, cpu_pairs=200000, cpu_budget=2.0),
                dict(name="C4-adaptive-mismatch5", n=20_000, length=10000, error=0.08, seed=datagen.SEEDS["C4"], cfg_kw=dict(C4, heuristic="adaptive", mismatch=5),
                     scheme="piggyback", survey_bytes=114e3 * 5 / 3, trim=50, cpu_pairs=60),
                # BiWFA (memory_mode="biwfa", SURVEY §8 f4: the O(s)-memory form the reference offers for long reads): bytes per pair = the packed
                # sequences + the op string (its rings live in the workspace and are re-used per score)
                dict(name="BiWFA-10kb", n=2000, length=10000, error=0.08, seed=datagen.SEEDS["C3"],
                     cfg_kw=dict(span="end-to-end", scope="full", memory_mode="biwfa"), scheme="none", survey_bytes=2 * 2500 + 10_800 + 8, cpu_pairs=16, cpu_budget=3.0),
                # (256 pairs: one per CU — the top levels of a 100 kb pair are one window per pair, 64 pairs left three quarters of the chip idle
                # and the line quoted 94 aln/s where README's 215 was measured on 256: VERDICT r05)
                dict(name="BiWFA-100kb", n=256, length=100000, error=0.08, seed=datagen.SEEDS["C5"],
                     cfg_kw=dict(span="end-to-end", scope="full", memory_mode="biwfa"), scheme="none", survey_bytes=2 * 25000 + 108_000 + 8, cpu_pairs=1, cpu_budget=3.0,
                     steps=1),
                dict(name="C3-explicit-history", n=100_000, length=10000, error=0.08, seed=datagen.SEEDS["C3"],
                     cfg_kw=dict(span="end-to-end", scope="full", heuristic="adaptive"), scheme="explicit", survey_bytes=750e3, env={"WFA_HIP_BAND_PB": "0"}),
                dict(name="C4-adaptive-explicit-history", n=20_000, length=10000, error=0.08, seed=datagen.SEEDS["C4"], cfg_kw=dict(C4, heuristic="adaptive"),
                     scheme="explicit", survey_bytes=750e3 * 5 / 3, trim=50, cpu_pairs=100, env={"WFA_HIP_BAND_PB": "0"}),
            ]
            for kw_ in plan:
                try:
                    xs.append(extra_config(**kw_))
                except Exception as e:   # the headline line is still printed, but the run ends non-zero (below)
                    xs.append({"name": kw_["name"], "error": repr(e)})
                    errors.append(f"{kw_['name']}: {e!r}")
            out["extra"]["configs"] = xs
            out["extra"]["configs_note"] = ("stated prefixes of the BASELINE streams: C1 at 1 M pairs (BASELINE names 1 k; and at 10 M, a batch that fills the chip), C3 100 k of 1 M, "
                                            "C4 as written (no heuristic) on 4 096 pairs (54 MB per pair: SURVEY's piggy-back figure) and with wf-adaptive on 100 k of "
                                            "1 M (its bytes per pair are C3's figure x 5/3 components); exact (no heuristic) gap-affine 10 kb on 8 192 pairs, score and full CIGAR "
                                            "(5.1 M M-offsets per pair: one history byte each); C5 with wf-adaptive on 8 192 of 100 k pairs (X-drop(20) / match = 0 as "
                                            "BASELINE writes C5 drops every pair after a few steps, SURVEY Q2); C2 above is the full 10 M.  Long reads keep the "
                                            "piggy-back history (one byte of origin codes per cell) in every memory mode; the *-explicit-history "
                                            "lines are the same configurations with WFA_HIP_BAND_PB=0")
            # the other half of the metric where a record that keeps only the scalars of `config` still shows it
            for x in xs:
                if "alignments_per_s" in x:
                    key = x["name"].lower().replace("-", "_").replace(".", "")
                    out["config"][f"{key}_alignments_per_s"] = x["alignments_per_s"]
                    out["config"][f"{key}_hbm_frac"] = x["roofline"]["frac"]
                    out["config"][f"{key}_parity_mismatches"] = x["parity_mismatches"]
                    if x["parity_mismatches"]:
                        errors.append(f"{x['name']}: {x['parity_mismatches']} parity mismatches of {x['parity_checked_pairs']}")
            try:   # what a pywfa user calls: WavefrontAligner.wavefront_align_batch(list of str) through the compiled host (pywfa_amd/host)
                out["extra"]["python_host"] = python_host_leg()
                out["config"]["python_batch_pairs_per_s"] = out["extra"]["python_host"]["pairs_per_s"]
                out["config"]["python_single_call_us"] = out["extra"]["python_host"]["single_call_us"]
            except Exception as e:
                errors.append(f"python host leg: {e!r}")
            try:   # C5 through the product's own multi-device entry (one process, every visible device)
                m = multi_leg(args.multi_pairs)
                out["extra"]["multi"] = m
                out["config"]["c5_multi_devices"] = m["devices"]
                for k in ("xdrop", "adaptive"):
                    out["config"][f"c5_multi_{k}_alignments_per_s"] = m[k]["alignments_per_s"]
                    if m[k]["violations"]:
                        errors.append(f"C5 multi {k}: {m[k]['violations']} violations")
            except Exception as e:
                errors.append(f"C5 multi leg: {e!r}")
        if c3 is not None:
            out["extra"]["c3_leg"] = c3
            out["config"]["c3_pairs_per_gpu"] = c3["pairs_per_gpu"]
            out["config"]["c3_leg_alignments_per_s" if n_gpus == 1 else "c3_alignments_per_s"] = c3["alignments_per_s"]
            out["config"]["c3_leg_hbm_frac" if n_gpus == 1 else "c3_hbm_frac"] = c3["hbm_frac"]
            out["config"]["c3_invalid_transcripts"] = c3["invalid_transcripts"]
            out["config"]["c3_rank_imbalance"] = c3["rank_imbalance"]
            if c3["invalid_transcripts"] or c3["completed"] != c3["pairs_per_gpu"] * n_gpus:
                errors.append(f"C3 leg: {c3['invalid_transcripts']} invalid transcripts, {c3['completed']} completed")
        cb = out.get("cpu_baseline") or {}
        if cb.get("parity_mismatches"):
            errors.append(f"C2: {cb['parity_mismatches']} parity mismatches of {cb['parity_checked_pairs']}")
        if (cb.get("all_threads") or {}).get("parity_mismatches"):
            errors.append(f"C2 (all threads): {cb['all_threads']['parity_mismatches']} parity mismatches")
        out["errors"] = errors
        out["detail_file"] = write_detail(out)
        # the LAST stdout line: one strict-JSON object well under the driver's 8 KB (tests/test_bench_line.py)
        print(format_line(compact_record(out)), flush=True)
    failed = bool(rank == 0 and out["errors"])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("bench.py: " + "; ".join(out["errors"]), file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
