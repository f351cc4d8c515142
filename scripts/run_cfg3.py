import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
for f in ("native", "torch"):
    out = bench.extra_cfg3(dev, 1 << 20, 5, f)
    print(f, round(out["ms_per_step"], 3), round(out["native_ms_per_step"], 3), out["loss"], {k: round(v["ms_per_launch"] * 1e3) for k, v in out["kernels"].items()})
