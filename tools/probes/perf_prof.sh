#!/bin/bash
# Development aid: per-kernel times of tools/gpu_perf.py configurations: bash tools/probes/perf_prof.sh E10 E5 ...
export TMPDIR=/tmp NO_CPU=1 BRIEF=1
for c in "$@"; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/pp_$c -o stats --output-format csv -- python3 tools/gpu_perf.py $c > gpurun_out/pp_$c.log 2>&1
  p=$(find gpurun_out/pp_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; grep "aln/s" gpurun_out/pp_$c.log
  python3 - "$p" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:105]:105s} calls={r['Calls']:>4s} total_us={float(r['TotalDurationNs'])/1e3:10.1f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
  rm -rf gpurun_out/pp_$c
done
