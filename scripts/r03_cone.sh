#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_tests.log
echo "== refill split"; timeout -k 10 300 python scripts/limit_sweep.py 1 4 16 0 2>/dev/null || exit 1
timeout -k 10 300 python scripts/testmode_iters.py | tail -2
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_CONE_WALK_SPLIT=1"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== walk split too"; timeout -k 10 300 python scripts/limit_sweep.py 0 2>/dev/null || exit 1
