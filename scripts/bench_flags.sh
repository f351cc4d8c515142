#!/bin/bash
# usage: [TILES="1024 2048"] scripts/bench_flags.sh "<flags1>" "<flags2>" ...
# rebuilds the library with each set of extra compiler flags and prints the bench's per-kernel times (A/B on ONE box:
# boxes differ by a few percent)
for f in "$@"; do
  export NERFACC_AMD_EXTRA_FLAGS="$f"
  python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  for t in ${TILES:-default}; do
    if [ "$t" = default ]; then unset NFA_SEG_TILE; else export NFA_SEG_TILE=$t; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('flags=[$f] tile=$t', round(d['ms_per_step'],4), round(d['headline_roofline']['frac'],4), d.get('parity_checked'), {k.replace('nfa_',''): round(v['ms_per_launch']*1e3,1) for k,v in d['kernels'].items() if v['ms_per_launch']>0.05})"
  done
done
unset NERFACC_AMD_EXTRA_FLAGS NFA_SEG_TILE
python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
