#!/bin/bash
# scratch: counters for one limited (test-mode) traversal call on cfg 5's scene; $1 = tag
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-lim}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/limit_sweep.py ${2:-4}"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/lim_${TAG}_1 -- $CMD > $OUT/lim_${TAG}_1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/lim_${TAG}_2 -- $CMD > $OUT/lim_${TAG}_2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/lim_${TAG}_3 -- $CMD > $OUT/lim_${TAG}_3.log 2>&1
python3 - <<PY
import csv, glob, collections
for i in (1, 2, 3):
    for f in glob.glob("$OUT/lim_${TAG}_%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "refill" not in k and "traverse_kernel" not in k and "cone_" not in k: continue
            acc[(k[:40], row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(k[:40], row["Counter_Name"])][1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{k:40s} {c:22s} per-launch {v / n:16.0f}  launches {n}")
PY
