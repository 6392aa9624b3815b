#!/usr/bin/env python3
"""profiles/traffic_c2.json from the counter passes of tools/gpu_counters.sh (development aid).

    python tools/make_traffic.py gpurun_out r02p        # reads gpurun_out/r02p_pmc_{fetch,write,sq1,sq2}, writes profiles/

HBM bytes of one C2 step = sum over the alignment kernels of (FETCH_SIZE x 2 [gfx950 correction, MI355X_MICROARCH.md HBM
section] + WRITE_SIZE), per dispatch; the on-chip figures of the dominant kernel go into "secondary".  The file is stamped
with the hash of the kernel sources it was measured on: bench.py uses it only when that hash matches the running build."""
import csv, glob, json, os, subprocess, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench

root, tag = sys.argv[1], sys.argv[2]
def measured_hash():
    """the hash the measured build printed on the GPU box (tools/gpu_counters.sh); this tree's if that file is absent"""
    f = os.path.join(root, f"{tag}_kernel_source_hash.txt")
    return open(f).read().strip() if os.path.exists(f) else bench.kernel_source_hash()
def per_dispatch(name):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob(os.path.join(root, f"{tag}_pmc_{name}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            acc[kn][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[kn].add(r["Dispatch_Id"])
    return {kn: {c: v / max(len(cnt[kn]), 1) for c, v in d.items()} for kn, d in acc.items()}, {kn: len(v) for kn, v in cnt.items()}

def is_align(kn):
    return any(x in kn for x in ("wfa_lane_kernel", "wfa_seg_kernel", "wfa_band_kernel", "wfa_general_kernel")) and "false, false>" not in kn.split("wfa_seg_kernel")[-1][:40] or "wfa_lane_kernel" in kn

fetch, nf = per_dispatch("fetch"); write, _ = per_dispatch("write"); sq1, _ = per_dispatch("sq1"); sq2, _ = per_dispatch("sq2")
# kernels of one step: every alignment kernel except the pilot's one-round segments (<..,16,false,false> etc. run once per batch)
step_kernels = [k for k in fetch if ("wfa_lane_kernel" in k or "true, false>" in k or "wfa_band_kernel" in k or "wfa_general_kernel" in k)]
fetch_kb = sum(fetch[k].get("FETCH_SIZE", 0.0) for k in step_kernels)
write_kb = sum(write.get(k, {}).get("WRITE_SIZE", 0.0) for k in step_kernels)
dom = max(step_kernels, key=lambda k: sq1.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0))
s1, s2 = sq1[dom], sq2[dom]
cycles = s1["GRBM_GUI_ACTIVE"] / 8.0              # summed over the 8 XCDs
simds = 256 * 4
# cost of a wave64 vector instruction per SIMD, measured by wall clock (tools/issue_rate.hip, 4 waves per SIMD): plain
# add / logic ops ("full rate") vs min / max / alignbit / cndmask / shifts / pk / dpp ("half rate")
def issue_ns(label):
    for line in open(os.path.join(root, f"{tag}_issue_rate.txt")):
        if line.startswith(label):
            part = line.split("w=4:")[1]
            return float(part.split("=")[1].split("ns")[0])
    return None
t_full, t_half = issue_ns("v_add_u32"), issue_ns("v_max_i32")
kernel_ns = None
for r in csv.DictReader(open(os.path.join(root, f"{tag}_kernel_stats.csv"))):
    if dom.split("(")[0] in r["Name"]:
        kernel_ns = float(r["AverageNs"])
# the kernel's DYNAMIC instruction mix (tools/lane_mix.py: static counts per region of the step x the region counters of the
# counting build; its total reproduces SQ_INSTS_VALU): the share of add / and / or / xor ("full rate"), of plain v_mov_b32 (priced
# at its own measured cost when the microbenchmark has it, otherwise at the half rate) — everything else at the half rate
MIX_FILE = next(f for f in (os.path.join(ROOT, "profiles", t + "_lane_mix.json") for t in (os.environ.get("WFA_PROFILE_TAG", "r05"), "r03")) if os.path.exists(f))
mix = json.load(open(MIX_FILE))
FULL_RATE_SHARE, MOV_SHARE = mix["full_rate_share"], mix["mov_share"]
t_mov = issue_ns("v_mov_b32 ") or t_half
valu = s2.get("SQ_INSTS_VALU", 0.0)
issue_ms = valu / simds * (FULL_RATE_SHARE * t_full + MOV_SHARE * t_mov + (1 - FULL_RATE_SHARE - MOV_SHARE) * t_half) * 1e-6
sec = {
    "kernel": dom.split("(")[0],
    "bound": "valu_issue",
    "what": "vector-issue time of the kernel's instruction stream = SQ_INSTS_VALU / 1024 SIMDs x the measured cost of a wave64 vector "
            "instruction on this chip (tools/issue_rate.hip, wall clock: full-rate add/logic ops and half-rate min/max/alignbit/cndmask/"
            "shift/pk/dpp ops, weighted by the kernel's DYNAMIC instruction mix: profiles/<tag>_lane_mix.json, tools/lane_mix.py), divided by the "
            "kernel's average duration (rocprofv3 --stats)",
    "frac": issue_ms / (kernel_ns * 1e-6),
    "valu_issue_ms": issue_ms, "kernel_ms": kernel_ns * 1e-6,
    "ns_per_full_rate_instr": t_full, "ns_per_half_rate_instr": t_half, "ns_per_v_mov": t_mov, "full_rate_share": FULL_RATE_SHARE, "v_mov_share": MOV_SHARE,
    "instruction_mix": {"file": os.path.relpath(MIX_FILE, ROOT), "dynamic_valu_total_model": mix["dynamic_valu_total"], "model_over_counter": mix.get("model_over_counter")},
    "frac_if_all_half_rate": valu / simds * t_half * 1e-6 / (kernel_ns * 1e-6), "frac_if_all_full_rate": valu / simds * t_full * 1e-6 / (kernel_ns * 1e-6),
    "valu_insts_per_dispatch": valu, "salu_insts_per_dispatch": s2.get("SQ_INSTS_SALU"), "lds_insts_per_dispatch": s2.get("SQ_INSTS_LDS"),
    "salu_insts_per_cycle_per_cu": s2.get("SQ_INSTS_SALU", 0.0) / (cycles * 256),
    "resident_waves_per_simd": s1["SQ_WAVE_CYCLES"] * 4.0 / (cycles * simds),
    "wave_cycles_split": {"issuing": s2.get("SQ_ACTIVE_INST_ANY", 0.0) / s1["SQ_WAVE_CYCLES"], "issue_stalled": s1["SQ_WAIT_INST_ANY"] / s1["SQ_WAVE_CYCLES"],
                          "waiting_on_counters": s2.get("SQ_WAIT_ANY", 0.0) / s1["SQ_WAVE_CYCLES"]},
    "gpu_cycles_per_dispatch": cycles,
}
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = None
out = {
    "workload": {"pairs_per_gpu": 10000000, "read_length": 150, "error": 0.02},
    "kernels": [k.split("(")[0] for k in step_kernels],
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
    "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE taken as is",
    "secondary": sec,
    "kernel_source_hash": measured_hash(), "git_head": head,
    "source": f"rocprofv3 --pmc, separate passes (tools/gpu_counters.sh {tag}): profiles/{tag}_pmc_*.txt; python3 bench.py --steps 2 --warmup 1",
}
import shutil
for part in ("kernel_stats.csv", "pmc_sq1.txt", "pmc_sq2.txt", "pmc_fetch.txt", "pmc_write.txt", "issue_rate.txt", "lane_regions.txt"):
    if os.path.exists(os.path.join(root, f"{tag}_{part}")):
        shutil.copy(os.path.join(root, f"{tag}_{part}"), os.path.join(ROOT, "profiles", f"{tag}_{part}"))
with open(os.path.join(ROOT, "profiles", "traffic_c2.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
