#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03_tests.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r03_tests.log
run() { timeout -k 10 300 python bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('step', round(d['ms_per_step'],4), 'headline', round(d['headline_roofline']['frac'],4), {n.replace('nfa_',''): round(v['ms_per_launch']*1e3,1) for n,v in k.items() if 'expand' in n or 'traverse' in n})"; }
echo "== staged 256 at a time"; run; run
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_EXP_RUNS_QMAX=1024"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== 1024 (as before)"; run; run
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_EXP_RUNS_QMAX=128"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== 128"; run; run
