#!/bin/bash
# refresh of the artefacts the test-mode changes touch: the two test-mode traces and the bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/r03_cfg5_testmode_trace $OUT/r03_cfg5_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_cfg5_testmode_trace -- python3 $R/bench.py --only cfg5_testmode --steps 12 > $OUT/r03_cfg5_testmode_trace.log 2>&1
echo "testmode rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_cfg5_trace -- python3 $R/bench.py --only cfg5 --steps 12 > $OUT/r03_cfg5_trace.log 2>&1
echo "cfg5 rc=$?"
cd $R && python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
