#!/bin/bash
run() { timeout -k 10 500 python bench.py --only $1 --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['$1']; d=d.get('test_mode_loop', d); print('$1', round(d['ms_per_image'],2), d['total_samples'])"; }
for f in 0.75 0.9 0.97 1.01 0.5; do export NERFACC_AMD_ALIVE_FRACTION=$f; echo "== alive list below $f"; run cfg5_testmode; run cfg2_testmode; done
