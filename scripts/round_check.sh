#!/bin/bash
# Run on the GPU box (via gpurun): the round's acceptance sequence -- full -m gpu tests, smoke(), the default bench line.
# Outputs: gpurun_out/<tag>_tests.log, <tag>_smoke.log, <tag>_bench.json / .err
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1 || { tail -30 $OUT/${TAG}_tests.log; exit 1; }
tail -3 $OUT/${TAG}_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1 || { tail -30 $OUT/${TAG}_smoke.log; exit 1; }
tail -2 $OUT/${TAG}_smoke.log
timeout -k 10 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -30 $OUT/${TAG}_bench.err; exit 1; }
python -c "
import json,sys
d=json.loads(open('$OUT/${TAG}_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','roofline')})
print('headline', d.get('headline_roofline'))
for k in ('cfg2_testmode','cfg4_per_rank','cfg3','cfg5'):
    v=d.get(k); print(k, json.dumps(v)[:700] if v else None)
"
