#!/bin/bash
# The headline loop (sequential and with the next batch's traversal prefetched) for a list of build variants, on one box:
#   scripts/r04_bench_ab.sh "<flags A>" "<flags B>" ...   ("-" = no extra flags)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in "$@"; do
  [ "$f" = "-" ] && f=""
  export NERFACC_AMD_EXTRA_FLAGS="$f"
  python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
  timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-extras --pipelined --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'flags': '$f', 'ms_per_step': round(d['ms_per_step'],4), 'pipelined_ms': round(d['pipelined']['ms_per_step'],4), 'walk_us': round(d['kernels']['nfa_traverse_runs']['ms_per_launch']*1e3,1), 'headline_frac': round(d['headline_roofline']['frac'],4)}))"
done
