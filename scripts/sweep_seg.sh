#!/bin/bash
# scratch: sweep tile size / prefetch for the segmented engine on the bench workload
for pipe in 0; do for tile in 512 1024 2048 4096; do  # the prefetch mode is a compile-time flag now (-DNFA_SEG_PIPE=0|1|2, scripts/bench_flags.sh)
  echo "== pipeline=$pipe tile=$tile"
  NFA_SEG_PIPELINE=$pipe NFA_SEG_TILE=$tile python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks=d['kernels']
print('  step %.3f ms native %.3f | ' % (d['ms_per_step'], d['native_ms_per_step']) + ' '.join('%s=%.0f' % (k.replace('nfa_','').replace('render_','r_'), v['ms_per_step']*1e3) for k,v in sorted(ks.items()) if 'render' in k or 'compact' in k))"
done; done
