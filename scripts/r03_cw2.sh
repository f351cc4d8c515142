#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cone or cfg5 or binned" > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r03_tests.log
timeout -k 10 500 python bench.py --only cfg5 --steps 18 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['cfg5']; print('cfg5 step', round(d['ms_per_step'],3), 'image', round(d['test_mode_loop']['ms_per_image'],2), d.get('parity_checked'), {k: round(v['ms_per_launch'],3) for k,v in d['kernels'].items() if 'cone' in k or 'traverse' in k})"
