#!/bin/bash
run() { timeout -k 10 500 python bench.py --only $1 --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['$1']; d=d.get('test_mode_loop', d); print('$1', round(d['ms_per_image'],2), d['total_samples'])"; }
echo "== rows compacted when < half alive (default)"; run cfg5_testmode; run cfg2_testmode
export NERFACC_AMD_TM_COMPACT_ROWS=0
echo "== never"; run cfg5_testmode; run cfg2_testmode
