#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_mode or cfg5" > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_tests.log
timeout -k 10 500 python bench.py --only cfg5_testmode --steps 12 > gpurun_out/r03_cfg5tm.json 2> gpurun_out/r03_cfg5tm.err
echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03_cfg5tm.json'))['cfg5_testmode']; print(d['test_mode_loop']); print({k:v for k,v in d.items() if 'parity' in k})"
