#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (development aid): python tools/pmc_summary.py <dir>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0][:120]
        acc[kn][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (kn, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[kn] += 1
for kn, d in acc.items():
    print(kn, "dispatches", cnt[kn])
    for c, v in sorted(d.items()): print(f"   {c:28s} total {v:.4g}  per dispatch {v / max(cnt[kn], 1):.4g}")
