#!/bin/bash
# SQ instruction / cycle counters of the kernels matching <pattern> for an arbitrary python command (run on the GPU box):
#   scripts/pmc_kernel.sh <tag> <kernel-name-pattern> <python script + args...>
TAG=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$R'); from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_sq1 -- python "$@" > $OUT/${TAG}_sq1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/${TAG}_sq2 -- python "$@" > $OUT/${TAG}_sq2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("${TAG}_sq1", "${TAG}_sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "$PAT" not in k: continue
            agg[k[:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
