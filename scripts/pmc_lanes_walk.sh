#!/bin/bash
# lane utilisation of the headline kernels (SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU))
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/lanes_walk
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/lanes_walk -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pipelined --no-extras > $OUT/lanes_walk.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/lanes_walk/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "nfa::" not in k: continue
        a = acc[k[:60]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, cs in acc.items():
        v = {c: a / n for c, (a, n) in cs.items()}
        if v.get("SQ_ACTIVE_INST_VALU", 0) > 0:
            print(f"{k:60s} waves {v['SQ_WAVES']:8.0f} VALU/wave {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:8.0f} SALU/wave {v['SQ_INSTS_SALU'] / v['SQ_WAVES']:8.0f} lane util {v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']):.3f}")
PY
