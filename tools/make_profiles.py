#!/usr/bin/env python3
"""profiles/<tag>_<cfg>_* and profiles/traffic_<cfg>.json from the passes of tools/gpu_profile_config.sh (development aid).

    python tools/make_profiles.py r03 C3 [bench-name]      # reads gpurun_out/r03_C3_*, writes profiles/

Kept under profiles/: the rocprofv3 --stats kernel summary, the per-kernel sums of the SQ / FETCH_SIZE / WRITE_SIZE passes, and a
JSON with what bench.py puts beside the configuration's roofline — HBM bytes per run (FETCH_SIZE x 2 [gfx950 correction,
MI355X_MICROARCH.md HBM section] + WRITE_SIZE over every kernel of a run) and the on-chip counters of the dominant kernel —
stamped with the hash of the kernel sources it was measured on (bench.py uses it only when that hash matches the running build)."""
import csv, json, os, re, shutil, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench

tag, cfg = sys.argv[1], sys.argv[2]
name = sys.argv[3] if len(sys.argv) > 3 else cfg
src = os.path.join(ROOT, "gpurun_out", f"{tag}_{cfg}")
def measured_hash():
    """the hash the measured build printed on the GPU box (tools/gpu_profile_config.sh); this tree's if that file is absent"""
    f = os.path.join(ROOT, "gpurun_out", f"{tag}_kernel_source_hash.txt")
    return open(f).read().strip() if os.path.exists(f) else bench.kernel_source_hash()
dst = os.path.join(ROOT, "profiles", f"{tag}_{cfg}")

def parse_pmc(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*) dispatches (\d+)$", line.rstrip("\n"))
        if m:
            cur = out.setdefault(m.group(1), {"dispatches": int(m.group(2))})
            continue
        m = re.match(r"^\s+(\S+)\s+total (\S+)\s+per dispatch (\S+)$", line.rstrip("\n"))
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out

for part in ("kernel_stats.csv", "pmc_sq1.txt", "pmc_sq2.txt", "pmc_fetch.txt", "pmc_write.txt"):
    shutil.copy(f"{src}_{part}", f"{dst}_{part}")
log = open(f"{src}_stats.log").read()
m = re.search(r"kernel_ms=\s*([\d.]+) aln/s=(\S+) .*pairs=(\d+) runs=(\d+)", log)
kernel_ms, rate, pairs, runs = float(m.group(1)), float(m.group(2)), int(m.group(3)), int(m.group(4))
sq1, sq2 = parse_pmc(f"{src}_pmc_sq1.txt"), parse_pmc(f"{src}_pmc_sq2.txt")
fetch, write = parse_pmc(f"{src}_pmc_fetch.txt"), parse_pmc(f"{src}_pmc_write.txt")
wfa = lambda k: "wfa::" in k and "wfa_pack_kernel" not in k and "wfa_repack2" not in k and "wfa_pilot" not in k
fetch_kb = sum(v.get("FETCH_SIZE", 0.0) for k, v in fetch.items() if wfa(k)) / runs
write_kb = sum(v.get("WRITE_SIZE", 0.0) for k, v in write.items() if wfa(k)) / runs
stats = []
for r in csv.DictReader(open(f"{src}_kernel_stats.csv")):
    if "wfa::" in r["Name"]:
        stats.append({"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6,
                      "total_ms_per_run": float(r["TotalDurationNs"]) * 1e-6 / runs})
stats.sort(key=lambda x: -x["total_ms_per_run"])
dom = stats[0]["kernel"]
def pick(d):
    for k, v in d.items():
        if k.startswith(dom[:118]):
            return v
    return {}
s1, s2 = pick(sq1), pick(sq2)
nd = max(s1.get("dispatches", 1), 1)
cycles = s1.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # summed over the 8 XCDs; total over the kernel's dispatches
simds, cus = 1024, 256
issue = {}
try:   # cost of a wave64 vector instruction per SIMD (tools/issue_rate.hip, profiles/r02_issue_rate.txt)
    for line in open(os.path.join(ROOT, "profiles", "r02_issue_rate.txt")):
        for lab in ("v_add_u32", "v_max_i32"):
            if line.startswith(lab):
                issue[lab] = float(line.split("w=4:")[1].split("=")[1].split("ns")[0])
except OSError:
    pass
valu, salu = s2.get("SQ_INSTS_VALU", 0.0), s2.get("SQ_INSTS_SALU", 0.0)
dom_ms = stats[0]["total_ms_per_run"] * runs           # all dispatches of the dominant kernel, like the counters
sec = {"kernel": dom, "dispatches_counted": nd,
       "valu_insts": valu, "salu_insts": salu, "lds_insts": s2.get("SQ_INSTS_LDS"), "waves": s2.get("SQ_WAVES"),
       "kernel_ms_total": dom_ms, "gpu_cycles": cycles,
       "valu_issue_frac_if_all_full_rate": (valu / simds * issue["v_add_u32"] * 1e-6 / dom_ms) if issue and dom_ms else None,
       "valu_issue_frac_if_all_half_rate": (valu / simds * issue["v_max_i32"] * 1e-6 / dom_ms) if issue and dom_ms else None,
       "salu_insts_per_cycle_per_cu": (salu / (cycles * cus)) if cycles else None,
       "resident_waves_per_simd": (s1.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / (cycles * simds)) if cycles else None,
       "wave_cycles_split": ({"issuing": s2.get("SQ_ACTIVE_INST_ANY", 0.0) / s1["SQ_WAVE_CYCLES"], "issue_stalled": s1.get("SQ_WAIT_INST_ANY", 0.0) / s1["SQ_WAVE_CYCLES"],
                              "waiting_on_counters": s2.get("SQ_WAIT_ANY", 0.0) / s1["SQ_WAVE_CYCLES"]} if s1.get("SQ_WAVE_CYCLES") else None),
       "what": "bounds of the vector-issue time of the dominant kernel: SQ_INSTS_VALU / 1024 SIMDs x the measured cost of a wave64 vector instruction "
               "(profiles/r02_issue_rate.txt: add / logic 'full rate', min / max / alignbit / cndmask / shifts / pk / dpp 'half rate') over its duration; "
               "the scalar unit issues at most one instruction per cycle per CU"}
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = None
out = {"config": name, "gpu_perf_config": cfg, "pairs": pairs, "runs_profiled": runs, "kernel_ms_per_run": kernel_ms, "alignments_per_s": rate,
       "kernels": stats, "FETCH_SIZE_KB_per_run": fetch_kb, "WRITE_SIZE_KB_per_run": write_kb,
       "hbm_bytes_per_run": int(fetch_kb * 1024 * 2 + write_kb * 1024), "hbm_bytes_per_pair": (fetch_kb * 1024 * 2 + write_kb * 1024) / pairs,
       "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE taken as is",
       "secondary": sec, "kernel_source_hash": measured_hash(), "git_head": head,
       "source": f"tools/gpu_profile_config.sh {tag} {cfg}: profiles/{tag}_{cfg}_kernel_stats.csv, profiles/{tag}_{cfg}_pmc_*.txt (separate rocprofv3 passes of python3 tools/gpu_perf.py {cfg})"}
with open(os.path.join(ROOT, "profiles", f"traffic_{name}.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1)[:3000])
