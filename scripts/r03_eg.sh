#!/bin/bash
run() { timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('step', round(d['ms_per_step'],4), {n.replace('nfa_',''): round(v['ms_per_launch']*1e3,1) for n,v in k.items() if 'expand' in n})"; }
echo "== default grid"; run
for g in 512 1024 2048 4096 8192; do echo "== NFA_EXP_GRID=$g"; NFA_EXP_GRID=$g run; done
