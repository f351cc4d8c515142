#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "headline", d["headline_roofline"]["frac"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel"])
print("pipelined", d.get("pipelined", {}).get("ms_per_step"), "parity", d.get("parity_checked"), d.get("parity"))
cb = d.get("cpu_baseline", {})
print("cpu", cb.get("value"), cb.get("cores"), cb.get("threads_speedup"), cb.get("agrees_with_pinned_composition"), cb.get("single_thread"))
for k in ("cfg2_compacting", "cfg2_random", "cfg3", "cfg5"):
    v = d.get(k, {})
    print(k, v.get("ms_per_step"), v.get("error"), v.get("parity_checked"), (v.get("cpu_baseline") or {}).get("value"), (v.get("test_mode_loop") or {}).get("ms_per_image"))
for k, v in d["kernels"].items():
    print("   ", k, round(v["ms_per_launch"] * 1e3, 1), "us")
PY
