#!/usr/bin/env python3
"""Development aid: the kernels of the LAST run of a rocprofv3 --kernel-trace (csv) as a timeline — start / end relative to the run's
first kernel, duration, stream/queue — so that what overlaps what (walks, the pipelined tail) can be read off.
python tools/trace_timeline.py <dir with *kernel_trace.csv> [gap_ms that separates runs, default 20]"""
import csv, glob, os, sys
d = sys.argv[1]
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
rows.sort()
# runs: split where the device was idle for more than `gap` ms
runs, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[0] - last_end > gap * 1e6:
        runs.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, r[1])
runs.append(cur)
run = max(runs[-3:], key=len) if len(runs) >= 3 else runs[-1]
run = runs[-1] if len(runs[-1]) > 3 else run
t0 = run[0][0]
print(f"{len(runs)} runs; the last one: {len(run)} kernels, {(max(r[1] for r in run) - t0) / 1e6:.3f} ms")
for s, e, name, q, g in run:
    short = name.split("(")[0][-90:]
    if (e - s) / 1e6 >= 0.05:
        print(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f}  {(e - s) / 1e6:8.3f} ms  q{q} grid {g:>9}  {short}")
