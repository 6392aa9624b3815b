#!/bin/bash
# Development aid: per-kernel times of tools/probes/laneh_probe.py for one (WHICH, E, FORCED)
export TMPDIR=/tmp
for f in ${FORCED_LIST:-1 2}; do
  FORCED=$f timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/lh_$f -o stats --output-format csv -- python3 tools/probes/laneh_probe.py > gpurun_out/lh_$f.log 2>&1
  p=$(find gpurun_out/lh_$f -name "*kernel_stats.csv" | head -1)
  echo "== forced $f"; grep LANE_HEUR gpurun_out/lh_$f.log
  python3 - "$p" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>4s} total_us={float(r['TotalDurationNs'])/1e3:10.1f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
  rm -rf gpurun_out/lh_$f
done
