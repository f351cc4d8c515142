#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/r03_cfg2_testmode_trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_cfg2_testmode_trace -- python3 $R/bench.py --only cfg2_testmode --steps 12 > $OUT/r03_cfg2_testmode_trace.log 2>&1
echo rc=$?
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/r03_cfg2_testmode_trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("GPU busy ms", tot / 1e6)
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
tail -1 $OUT/r03_cfg2_testmode_trace.log | cut -c1-300
