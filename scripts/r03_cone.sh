#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned or cone or cfg5_regime or automatic" > gpurun_out/r03_t4.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r03_t4.log
timeout -k 10 500 python bench.py --only cfg5 --steps 12 > gpurun_out/r03_cfg5.json 2> gpurun_out/r03_cfg5.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_cfg5.json"))["cfg5"]
print("ms_per_step", d["ms_per_step"], "parity", d.get("parity_checked"), d.get("parity_error"))
for k, v in d["kernels"].items():
    print("   ", k, round(v["ms_per_launch"], 3), "ms")
PY
