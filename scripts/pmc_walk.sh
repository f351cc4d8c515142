#!/bin/bash
# SQ instruction / cycle counters of the walk kernel alone (run on the GPU box via gpurun): scripts/pmc_walk.sh <tag> [extra bench args]
TAG=${1:-walk}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$R'); from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
CMD="python $R/scripts/walk_bench.py --reps 3 $@"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_sq1 -- $CMD > $OUT/${TAG}_sq1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/${TAG}_sq2 -- $CMD > $OUT/${TAG}_sq2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("${TAG}_sq1", "${TAG}_sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "walk_kernel" not in k: continue
            agg[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, {c: sum(x) / len(x) for c, x in v.items()})
PY
