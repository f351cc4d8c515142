#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r03_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_trace -- python3 scripts/onepass_trace.py 128 > gpurun_out/r03_trace.log 2>&1
echo rc=$?
f=$(find gpurun_out/r03_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-30:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(f'{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} us  q={r.get("Queue_Id","?")}  {r["Kernel_Name"][:70]}')
PY
