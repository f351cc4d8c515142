#!/bin/bash
# scratch: one bench run, compact per-kernel summary
python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rays/s %.1fM  ms/step %.3f  native %.3f ms  M=%d  pipelined %s' % (d['value']/1e6, d['ms_per_step'], d['native_ms_per_step'], d['config']['samples_before_compaction'], ('%.1fM (%.3f ms)' % (d['pipelined']['value']/1e6, d['pipelined']['ms_per_step'])) if 'pipelined' in d else '-'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step']):
    print('  %-34s %7.1f us x%.0f %8.1f GB/s' % (k, v['ms_per_step']*1e3, v['launches_per_step'], v.get('achieved_GBps',0)))"
