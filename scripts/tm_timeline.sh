#!/bin/bash
# kernel timeline of one iteration of the cfg-2 test-mode loop: per-kernel start/end and the idle gaps between consecutive kernels
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r03_tmtl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_tmtl -- python3 bench.py --only cfg2_testmode --steps 8 > gpurun_out/r03_tmtl.log 2>&1
echo rc=$?
f=$(find gpurun_out/r03_tmtl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "walk_kernel" in r["Kernel_Name"]]
a, b = idx[-12], idx[-11]      # an iteration in the middle of the last image
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += (e - s)
    print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {r["Kernel_Name"][:90]}')
    prev_end = max(prev_end or e, e)
print("iteration wall us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "busy us", busy / 1e3, "kernels", b - a)
PY
