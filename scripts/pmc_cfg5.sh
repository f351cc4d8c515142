#!/bin/bash
# scratch: counters for the cone-angle traversal kernels on cfg 5 (run on the GPU box via gpurun); $1 = tag
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-walk}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/bench_configs.py cfg5 --steps 2 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/c5_${TAG}_1 -- $CMD > $OUT/c5_${TAG}_1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/c5_${TAG}_2 -- $CMD > $OUT/c5_${TAG}_2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/c5_${TAG}_3 -- $CMD > $OUT/c5_${TAG}_3.log 2>&1
echo done $TAG
