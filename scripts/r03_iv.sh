#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_tests.log
timeout -k 10 300 python scripts/api_traverse_probe.py 2>/dev/null || exit 1
