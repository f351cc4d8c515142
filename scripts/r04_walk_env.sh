#!/bin/bash
# walk_bench under a list of environment settings (one build): scripts/r04_walk_env.sh "VAR=val" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
for e in "$@"; do
  for res in 128 256; do
    env $e timeout -k 10 200 python $R/scripts/walk_bench.py --res $res --tag="$e" 2>/dev/null
  done
done
