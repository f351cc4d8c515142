#!/bin/bash
# SQ instruction / cycle counters of every nfa:: kernel of the bench step (run on the GPU box): scripts/pmc_step.sh <tag>
# -> gpurun_out/<tag>_sq.json (per kernel: waves, VALU / SALU / LDS / VMEM instructions, busy and wait cycles)
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pipelined --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_sqa -- $CMD > $OUT/${TAG}_sqa.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/${TAG}_sqb -- $CMD > $OUT/${TAG}_sqb.log 2>&1
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("${TAG}_sqa", "${TAG}_sqb"):
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "nfa::" not in k: continue
            k = k.replace("void ", "").split("(")[0][:90]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items()}
json.dump(out, open("$OUT/${TAG}_sq.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:8]:
    w = v.get("SQ_WAVES", 1)
    print(k[:60], "waves", int(w), "VALU/wave", round(v.get("SQ_INSTS_VALU", 0) / w), "SALU/wave", round(v.get("SQ_INSTS_SALU", 0) / w),
          "LDS/wave", round(v.get("SQ_INSTS_LDS", 0) / w), "VMEM/wave", round((v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / w),
          "valu_busy", round(v.get("SQ_ACTIVE_INST_VALU", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3), "wait", round(v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3))
PY
