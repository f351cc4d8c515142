#!/bin/bash
run() { timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('step', round(d['ms_per_step'],4), 'pipelined', round(d['pipelined']['ms_per_step'],4), {n.replace('nfa_',''): round(v['ms_per_launch']*1e3,1) for n,v in k.items() if 'cumsum' not in n and 'tiles' not in n})"; }
echo "== default"; run
for t in 512 2048 4096; do echo "== NFA_SEG_TILE=$t"; NFA_SEG_TILE=$t run; done
echo "== no speculation"; NERFACC_AMD_SPECULATE=0 run
echo "== default again"; run
