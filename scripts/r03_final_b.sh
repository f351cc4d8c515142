#!/bin/bash
bash scripts/collect_profiles.sh r03 2>&1 | tail -2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_cfg5_testmode_trace -- python3 $R/bench.py --only cfg5_testmode --steps 12 > $OUT/r03_cfg5_testmode_trace.log 2>&1
echo "testmode rc=$?"
cd $R && python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
