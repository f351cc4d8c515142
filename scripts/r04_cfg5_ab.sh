#!/bin/bash
# cfg 5's step for a list of build variants on one box: scripts/r04_cfg5_ab.sh "<flags A>" "<flags B>" ...   ("-" = none)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in "$@"; do
  [ "$f" = "-" ] && f=""
  export NERFACC_AMD_EXTRA_FLAGS="$f"
  python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
  timeout -k 10 400 python $R/bench.py --only cfg5 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d.get('cfg5', d)
print(json.dumps({'flags': '$f', 'ms_per_step': round(c['ms_per_step'],3), 'cone_walk_ms': round(c['kernels']['nfa_traverse_cone_walk']['ms_per_launch'],3), 'test_mode_ms': round(c['test_mode_loop']['ms_per_image'],1), 'parity': c.get('parity_checked')}))"
done
