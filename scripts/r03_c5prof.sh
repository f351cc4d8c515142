#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/r03_cfg5_trace
NFA_BENCH_CFG5_TRAIN_ONLY=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_cfg5_trace -- python3 $R/bench.py --only cfg5 --steps 12 > $OUT/r03_cfg5_trace.log 2>&1
echo rc=$?
f=$(find $OUT/r03_cfg5_trace -name "*kernel_stats.csv" | head -1)
head -24 $f | cut -c1-150
tail -2 $OUT/r03_cfg5_trace.log | cut -c1-600
