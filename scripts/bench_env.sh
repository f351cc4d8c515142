#!/bin/bash
# usage: scripts/bench_env.sh "VAR=val VAR2=val" ...   -- one bench run per environment setting (A/B on one box)
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('env=[$e]', round(d['ms_per_step'],4), round(d['headline_roofline']['frac'],4), {k.replace('nfa_',''): round(v['ms_per_launch']*1e3,1) for k,v in d['kernels'].items() if v['ms_per_launch']>0.05})"
done
