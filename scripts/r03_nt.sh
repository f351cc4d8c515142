#!/bin/bash
run() { timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('step', round(d['ms_per_step'],4), 'pipelined', round(d['pipelined']['ms_per_step'],4), 'headline', round(d['headline_roofline']['frac'],4), {n: round(v['ms_per_launch']*1e3,1) for n,v in k.items()})"; }
echo "== NT stores (default)"; run; run
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_NT_EXPAND=0"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== plain stores"; run; run
