#!/bin/bash
# kernel timeline of the headline step: per-kernel start/end and the idle gaps between consecutive kernels
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r04_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-pipelined --no-extras > gpurun_out/r04_tl.log 2>&1
echo rc=$?
f=$(find gpurun_out/r04_tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: from the last walk kernel
idx = [i for i, r in enumerate(rows) if ("walk_kernel" in r["Kernel_Name"] or "walk_lattice_kernel" in r["Kernel_Name"])]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += (e - s)
    print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {r["Kernel_Name"][:80]}')
    prev_end = max(prev_end or e, e)
print("step wall us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "busy us", busy / 1e3)
PY
