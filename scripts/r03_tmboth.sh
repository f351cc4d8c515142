#!/bin/bash
run() { timeout -k 10 500 python bench.py --only $1 --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['$1']; d=d.get('test_mode_loop', d); print('$1', round(d['ms_per_image'],2), d['total_samples'])"; }
run cfg2_testmode; run cfg2_testmode; run cfg5_testmode
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'pipelined', round(d['pipelined']['ms_per_step'],4))"
