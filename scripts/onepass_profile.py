"""Where the waves of nfa_traverse_onepass spend their time (library built with NERFACC_AMD_EXTRA_FLAGS=-DNFA_OP_PROFILE)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from nerfacc_amd import grid as G

dev = torch.device("cuda:0")
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = bench.make_workload(dev, 1 << 20, res, "shell10", "image")
est = w["estimator"]
n = 1 << 20
near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
prof = torch.zeros(10 * 4 * 4096, dtype=torch.int64, device=dev)
os.environ["NFA_OP_PROFILE_PTR"] = str(prof.data_ptr())
args = (w["rays_o"], w["rays_d"], est.binaries, est.aabbs, near, far, w["step"], 0.0)
for it in range(4):
    prof.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = G._traverse_samples(*args, near_hint=0.0)
    e1.record()
    torch.cuda.synchronize()
    print("iter", it, "samples", out[0].numel(), "ms", e0.elapsed_time(e1))
p = prof.cpu().numpy().reshape(-1, 10)
p = p[p[:, 9] > 0]
print("waves", len(p))
t0 = p[:, 8].min()
span = (p[:, 9].max() - t0) / 100.0  # wall_clock64: 100 MHz -> us
print("kernel span us", span, "first->last start us", (p[:, 8].max() - t0) / 100.0)
names = ["ticket", "resolve", "header", "chunks", "units", "-"]
for i, nm in enumerate(names):
    print(f"{nm:14s} mean {p[:, i].mean():12.1f}  max {p[:, i].max():12d}  sum {p[:, i].sum():14d}")
cyc = p[:, 0] + p[:, 1] + p[:, 2] + p[:, 3]
life = (p[:, 9] - p[:, 8]) / 100.0
print("wave life us mean", life.mean(), "max", life.max(), "; accounted cycles mean", cyc.mean(), "-> MHz", cyc.mean() / life.mean())
print("end times us percentiles", np.percentile((p[:, 9] - t0) / 100.0, [1, 10, 50, 90, 99, 100]))
