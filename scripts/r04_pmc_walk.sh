#!/bin/bash
# SQ counters of walk_kernel for a list of build variants: scripts/r04_pmc_walk.sh <tag> "<flags>" [walk_bench args]
TAG=$1; FLAGS=$2; shift; shift
[ "$FLAGS" = "-" ] && FLAGS=""
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export NERFACC_AMD_EXTRA_FLAGS="$FLAGS"
python -c "import sys; sys.path.insert(0, '$R'); from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/walk_bench.py --reps 3 $@"
rm -rf $OUT/${TAG}_sq1 $OUT/${TAG}_sq2 $OUT/${TAG}_sq3
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_sq1 -- $CMD > $OUT/${TAG}_sq1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/${TAG}_sq2 -- $CMD > $OUT/${TAG}_sq2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/${TAG}_sq3 -- $CMD > $OUT/${TAG}_sq3.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = {}
for d in ("${TAG}_sq1", "${TAG}_sq2", "${TAG}_sq3"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "walk_lattice_kernel" not in row["Kernel_Name"] and "walk_kernel" not in row["Kernel_Name"]: continue
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, x in agg.items(): tot[c] = sum(x) / len(x)
w = tot.get("SQ_WAVES", 1)
print("${TAG} [${FLAGS}]", {c: (round(v / w, 1) if c != "SQ_WAVES" else v) for c, v in sorted(tot.items())})
if "SQ_THREAD_CYCLES_VALU" in tot: print("  lane util", round(tot["SQ_THREAD_CYCLES_VALU"] / (64 * tot["SQ_ACTIVE_INST_VALU"]), 3))
PY
