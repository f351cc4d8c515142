#!/bin/bash
run() { timeout -k 10 300 python bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('step', round(d['ms_per_step'],4), {n.replace('nfa_',''): round(v['ms_per_launch']*1e3,1) for n,v in k.items() if 'fused' in n or 'visib' in n})"; }
echo "== default"; run; run
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_SEG_OCC_HINTS=1"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== occupancy hints"; run; run
