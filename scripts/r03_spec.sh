#!/bin/bash
run() { timeout -k 10 500 python bench.py --only $1 --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['$1']; t=d.get('test_mode_loop', d); print('$1', round(t['ms_per_image'],2), t['total_samples'])"; }
echo "== speculative expansion"; run cfg2_testmode; run cfg2_testmode
export NERFACC_AMD_TM_SPECULATE=0
echo "== plain read"; run cfg2_testmode; run cfg2_testmode
