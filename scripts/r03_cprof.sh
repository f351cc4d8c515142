#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cone or cfg5 or fuzz or test_mode" > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_tests.log
echo "== staged"; timeout -k 10 300 python scripts/limit_sweep.py 1 4 16 2>/dev/null || exit 1
echo "== lists from memory"; NFA_CONE_STAGED=0 timeout -k 10 300 python scripts/limit_sweep.py 1 4 16 2>/dev/null || exit 1
timeout -k 10 300 python scripts/testmode_iters.py | tail -1
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_CONE_PROFILE"
python -c "from nerfacc_amd import _build; _build.build(force=True)" 2>&1 | tail -1
timeout -k 10 300 python scripts/cone_profile.py
