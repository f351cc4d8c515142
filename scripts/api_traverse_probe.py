"""The API-faithful traverse_grids on BASELINE cfg 2 (intervals + samples): per-call HIP-event times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as na
from nerfacc_amd import _backend as B

dev = torch.device("cuda:0")
w = bench.make_workload(dev, 1 << 20, 128, "shell10", "image")
est = w["estimator"]
rec = []
orig = B.call


def timed(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *a); e1.record()
    rec.append((name, e0, e1))


for it in range(5):
    if it == 2:
        B.call = timed
    iv, sm, term = na.traverse_grids(w["rays_o"], w["rays_d"], est.binaries, est.aabbs, step_size=w["step"])
torch.cuda.synchronize()
B.call = orig
agg = {}
for name, e0, e1 in rec:
    agg.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
print("edges", iv.vals.numel(), "samples", sm.vals.numel())
for k, v in agg.items():
    print(f"{k:40s} {min(v):8.1f} us")
