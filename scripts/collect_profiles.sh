#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM byte counters for the bench command.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass; no --pmc together
# with trace domains other than kernel-trace).  Outputs land in gpurun_out/<tag>_*; summarise with
# scripts/summarize_profiles.py, which writes the files committed under profiles/.
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
(cd $R && python -c "from nerfacc_amd import _build; _build.build(); print(_build._source_hash())" > $OUT/${TAG}_source_hash.txt)   # ties the counters to the library they were taken from
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-pipelined --no-extras"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $CMD > $OUT/${TAG}_trace.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- $CMD > $OUT/${TAG}_write.log 2>&1
# secondary configurations (bench.py --only): kernel-trace stats each
for cfg in cfg2_compacting cfg2_random cfg3 cfg5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${cfg}_trace -- python $R/bench.py --only $cfg --steps 12 > $OUT/${TAG}_${cfg}_trace.log 2>&1
done
# cfg 4's grid (256^3 shell10, the workload of bench.py --gpus N > 1) on one GPU
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg4_256_trace -- python $R/bench.py --res 256 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-pipelined --no-extras > $OUT/${TAG}_cfg4_256_trace.log 2>&1
echo "profiles collected for $TAG"
