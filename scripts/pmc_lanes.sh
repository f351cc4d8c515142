#!/bin/bash
# lane utilisation of the unlimited cone walk on cfg 5's scene, rays binned by crossed cells or not
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for b in 0 1; do
export NFA_LS_BIN=$b
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/lanes_$b -- python3 $R/scripts/limit_sweep.py 0 > $OUT/lanes_$b.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/lanes_$b/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "cone_walk_kernel" not in k: continue
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
    v = {c: a / n for c, (a, n) in acc.items()}
    print("bin_rays=$b", {c: round(x) for c, x in v.items()}, "lane utilisation", round(v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), 3) if "SQ_THREAD_CYCLES_VALU" in v else None)
PY
grep "limit" $OUT/lanes_$b.log
done
