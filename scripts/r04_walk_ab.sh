#!/bin/bash
# A/B of walk.hip build variants on one box: scripts/r04_walk_ab.sh "<flags A>" "<flags B>" ...   ("-" = no extra flags)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in "$@"; do
  [ "$f" = "-" ] && f=""
  export NERFACC_AMD_EXTRA_FLAGS="$f"
  python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
  for res in 128 256; do
    timeout -k 10 200 python $R/scripts/walk_bench.py --res $res --tag="$f" 2>/dev/null
  done
done
