#!/usr/bin/env python3
"""Dynamic instruction mix of wfa_lane_kernel<2,4,1,false,false> (the C2 hot loop): which share of its vector instructions runs at
the full issue rate (development aid).

    python tools/lane_mix.py [counts-file]

1. The kernel is compiled for gfx950 with region marks (-DWFA_LANE_REGION_MARKS=1: comments in the assembly around the refill,
   the first-probe block of a register, a round of a second run, the parked-run loop and its write-back, the hand-over block),
   and the vector instructions of every region are counted by class: "full rate" = v_add_u32 / v_and / v_or / v_xor (0.91-0.93 ns
   per wave64 instruction and SIMD, profiles/r02_issue_rate.txt), everything else "half rate" (min / max / alignbit / cndmask /
   shifts / sub / add3 / pk / dpp / ffbl / cmp / mov-with-dpp: 1.64-1.83 ns); v_mov_b32 and v_readlane etc. are listed apart.
2. How often each region runs comes from the counting build of the same kernel on the C2 batch (-DWFA_LANE_DEBUG_COUNTERS=1,
   WFA_HIP_LANE_DEBUG=1; the line is kept in profiles/r03_lane_regions.txt): wave-steps, refills, probe blocks, second-run
   rounds, parked runs and rounds, hand-over blocks.
3. Dynamic count of a class = sum over regions of executions x static count; what lies outside the marks inside the main loop is
   straight-line per-step code (tests, termination, compute-next).  The total is checked against SQ_INSTS_VALU of the counter pass.
Writes profiles/<tag>_lane_mix.json (tag: WFA_PROFILE_TAG, default r05) (tools/make_traffic.py takes full_rate_share from it)."""
import json, os, re, subprocess, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "pywfa_amd", "csrc")
ASM = "/tmp/k_lane_marks.s"
KERNEL = "_ZN3wfa15wfa_lane_kernelILi2ELi4ELi1ELb0ELb0ELi0ELi8EEEvNS_8FastArgsEii"   # (round 6: the LIN and NRP parameters)
FULL_RATE = ("v_add_u32", "v_add_co_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_and_or_b32", "v_or3_b32", "v_not_b32")


def classify(op):
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "half"
    base = op.replace("_e32", "").replace("_e64", "")
    if base in FULL_RATE:
        return "full"
    if base in ("v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"):
        return "mov"
    if base.startswith("v_readlane") or base.startswith("v_readfirstlane") or base.startswith("v_writelane"):
        return "lane"
    return "half"


def compile_marks():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-DWFA_TU_INDEX=0", "-DWFA_LANE_REGION_MARKS=1", "--cuda-device-only", "-S", os.path.join(CSRC, "k_lane.hip"), "-o", ASM]
    subprocess.run(cmd, check=True, capture_output=True)


def static_counts():
    regions = collections.defaultdict(lambda: collections.Counter())
    copies = collections.Counter()
    inside, stack, loop_seen = False, [], False
    first_loop_label = None
    for line in open(ASM):
        t = line.strip()
        if t.startswith(KERNEL + ":"):
            inside = True
            continue
        if inside and t.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"; WFA_MARK (\w+)_(begin|end)", t)
        if m:
            name, kind = m.group(1), m.group(2)
            if name == "looptop":
                loop_seen = True
                continue
            if kind == "begin":
                stack.append(name); copies[name] += 1
            elif stack and stack[-1] == name:
                stack.pop()
            continue
        m = re.match(r"(v_[a-z0-9_]+)\s", t)
        if not m:
            continue
        region = stack[-1] if stack else ("step" if loop_seen else "prologue")
        regions[region][classify(m.group(1))] += 1
        regions[region]["ops:" + m.group(1)] += 1
    return regions, copies


def main():
    compile_marks()
    regions, copies = static_counts()
    TAG = os.environ.get("WFA_PROFILE_TAG", "r05")
    counts_file = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", TAG + "_lane_regions.txt")
    txt = open(counts_file).read()
    m = re.search(r"(\d+) wave-steps, (\d+) refills, (\d+) parked runs, (\d+) parked rounds, (\d+) probe blocks, (\d+) second-run rounds, (\d+) hand-over blocks, (\d+) second runs", txt)
    steps, refills, parked, prounds, probes, now_rounds, bd, now_runs = [int(x) for x in m.groups()]
    # executions of ONE copy of each region (the unrolled copies of a region are equal: their static counts are averaged)
    execs = {"step": steps, "refill": refills, "probe": probes, "nowfix": now_runs, "now": now_rounds, "parkfix": parked, "parked": prounds, "bd": bd}
    dyn = collections.Counter()
    per_region = {}
    for name, cnt in regions.items():
        n_copies = max(copies.get(name, 1), 1)
        e = execs.get(name, 0)
        per_region[name] = {"copies": n_copies, "executions": e,
                            "static_per_copy": {k: v / n_copies for k, v in cnt.items() if not k.startswith("ops:")}}
        for k, v in cnt.items():
            if not k.startswith("ops:"):
                dyn[k] += e * v / n_copies
    # (the marks nest: probe contains now; parkfix contains parked — static counts are attributed to the innermost region)
    total = sum(dyn.values())
    out = {"kernel": "wfa_lane_kernel<2,4,1,false>", "counts_file": os.path.relpath(counts_file, ROOT),
           "region_executions": execs, "regions": per_region,
           "dynamic_valu_by_class": dict(dyn), "dynamic_valu_total": total,
           "full_rate_share": dyn["full"] / total, "mov_share": dyn["mov"] / total, "lane_share": dyn["lane"] / total,
           "full_rate_ops": list(FULL_RATE),
           "note": "v_mov_b32 / v_readlane are counted at the half rate in the roofline (they are none of the measured full-rate ops)"}
    try:
        import glob
        for f in glob.glob(os.path.join(ROOT, "profiles", TAG + "_pmc_sq2.txt")) + glob.glob(os.path.join(ROOT, "profiles", "r03_pmc_sq2.txt")):
            cur = None
            for line in open(f):
                if "wfa_lane_kernel<2, 4, 1" in line:
                    cur = int(line.split("dispatches")[1])
                elif cur and "SQ_INSTS_VALU" in line:
                    out["sq_insts_valu_per_dispatch"] = float(line.split("per dispatch")[1])
                    out["sq_insts_valu_file"] = os.path.relpath(f, ROOT)
                    cur = None
            if "sq_insts_valu_per_dispatch" in out:
                break
        if "sq_insts_valu_per_dispatch" in out:
            out["model_over_counter"] = total / out["sq_insts_valu_per_dispatch"]
    except OSError:
        pass
    with open(os.path.join(ROOT, "profiles", TAG + "_lane_mix.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "regions"}, indent=1))


if __name__ == "__main__":
    main()
