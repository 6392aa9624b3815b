#!/bin/bash
# Development aid: per-kernel times of the short-read heuristic configurations (tools/probes/short_heur.py) — one rocprofv3 run each.
export TMPDIR=/tmp NO_CPU=1 BRIEF=1
for w in ${WHICH_LIST:-adapt xdrop match1 wild}; do
  WHICH=$w timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/sh_$w -o stats --output-format csv -- python3 tools/probes/short_heur.py > gpurun_out/sh_$w.log 2>&1
  f=$(find gpurun_out/sh_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; tail -2 gpurun_out/sh_$w.log
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>4s} total_us={float(r['TotalDurationNs'])/1e3:10.1f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
  rm -rf gpurun_out/sh_$w
done
