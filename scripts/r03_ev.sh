#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_tests.log
timeout -k 10 300 python scripts/limit_sweep.py 1 4 16 0 2>/dev/null || exit 1
timeout -k 10 300 python scripts/testmode_iters.py | tail -2
timeout -k 10 500 python bench.py --only cfg5_testmode --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['cfg5_testmode']['test_mode_loop']; print(round(d['ms_per_image'],2), d['total_samples'])"
