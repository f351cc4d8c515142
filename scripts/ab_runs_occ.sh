#!/bin/bash
# scratch: occupancy / event-list-size variants of the run-length walk on the bench workload
for v in "-DNFA_RUNS_WAVES=5 -DNFA_EV_MAX=28" "-DNFA_RUNS_WAVES=5 -DNFA_EV_MAX=24" "-DNFA_RUNS_WAVES=5 -DNFA_EV_MAX=20" "-DNFA_EV_MAX=24" "-DNFA_RUNS_WAVES=6 -DNFA_EV_MAX=24"; do
  export NERFACC_AMD_EXTRA_FLAGS="$v"   # the import-time staleness check hashes the flags too
  python nerfacc_amd/_build.py > /dev/null 2>&1 || exit 1
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$v', '| step %.3f ms | runs %.0f us expand %.0f us' % (d['ms_per_step'], k['nfa_traverse_runs']['ms_per_step']*1e3, k['nfa_expand_runs']['ms_per_step']*1e3))" || exit 1
done
unset NERFACC_AMD_EXTRA_FLAGS
python nerfacc_amd/_build.py > /dev/null 2>&1
