#!/bin/bash
# kernel trace of one mid-size full-CIGAR resident run (development aid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/mbf.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tests")
import common
from pywfa_amd import datagen, _native
oc, nc = common.configs_pair(span="end-to-end", scope="full")
al = _native.Aligner(nc)
batch = datagen.generate(65536, 150, 0.02, 5)
rb = al.batch(batch)
for _ in range(6): rb.run(); rb.sync()
PY
rocprofv3 --kernel-trace -d gpurun_out/mbprof -o mb --output-format csv -- python3 /tmp/mbf.py > gpurun_out/mbprof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/mbprof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run: kernels after the last lane kernel launch
idx = max(i for i, r in enumerate(rows) if "wfa_lane_kernel" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  {r["Kernel_Name"][:90]}')
PY
