#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
run() { timeout -k 10 300 python $R/bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'pipelined', round(d['pipelined']['ms_per_step'],4))"; }
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stall_trace_$i -- python3 $R/bench.py --only cfg5_testmode --steps 12 > /tmp/stall_$i.log 2>&1
echo "after rocprof run $i:"; run; run; run
done
