#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pdf_loss or propnet or captured" > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03_tests.log
run() { timeout -k 10 400 python bench.py --only cfg3 --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['cfg3']; print(round(d['ms_per_step'],4))"; }
echo "== mean form"; run; run
echo "== loss array"; NERFACC_AMD_FUSE_LOSS_MEAN=0 run; NERFACC_AMD_FUSE_LOSS_MEAN=0 run
