#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03_smoke.log
SECONDS=0
bash scripts/r03_bench.sh
echo "bench wall ${SECONDS}s"
