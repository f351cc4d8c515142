#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the bench step on unrelated rays, without and with ray binning.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in off on; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r01d_random_$mode -o rnd -- python $R/bench.py --ray-variant random --bin-rays $mode --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-pipelined > $OUT/r01d_random_$mode.log 2>&1 || exit 1
done
echo collected
