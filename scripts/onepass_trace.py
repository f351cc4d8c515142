"""A few one-pass traversals of the bench workload (for rocprofv3 --kernel-trace: start / end of the walk and the expander)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfacc_amd import grid as G

dev = torch.device("cuda:0")
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = bench.make_workload(dev, 1 << 20, res, "shell10", "image")
est = w["estimator"]
n = 1 << 20
near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
args = (w["rays_o"], w["rays_d"], est.binaries, est.aabbs, near, far, w["step"], 0.0)
for it in range(6):
    out = G._traverse_samples(*args, near_hint=0.0)
    torch.cuda.synchronize()
print("samples", out[0].numel())
