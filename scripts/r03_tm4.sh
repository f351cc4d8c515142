#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_mode or cfg5 or marching or alive" > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_tests.log
run() { timeout -k 10 500 python bench.py --only $1 --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['$1']; t=d.get('test_mode_loop', d); print('$1', round(t['ms_per_image'],2), t['total_samples'], d.get('parity', d.get('test_mode_parity')))"; }
run cfg2_testmode; run cfg2_testmode; run cfg5_testmode
