"""Host-side cost of one limited traversal call (test-mode iteration) at a size where the GPU work is negligible."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerfacc_amd as na
from nerfacc_amd import grid as GR

dev = torch.device("cuda:0")
res, G, R = 128, 4, 1 << 16
est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=G).to(dev)
g = torch.Generator(device=dev); g.manual_seed(5)
est.binaries = torch.rand((G, res, res, res), device=dev, generator=g) < 0.02
rng = np.random.default_rng(5)
o = torch.from_numpy(rng.random((R, 3)).astype(np.float32) - 0.5).to(dev)
d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
d = torch.from_numpy(d).to(dev)
near = torch.full((R,), 0.2, device=dev); far = torch.full((R,), 1e10, device=dev)
mask = torch.rand(R, device=dev) < 0.5
n_alive = int(mask.sum())
tmin, tmax, hits = na.ray_aabb_intersect(o, d, est.aabbs)
ts, ti = torch.sort(torch.cat([tmin, tmax], -1), -1)


def call():
    return GR._traverse_samples(o, d, est.binaries, est.aabbs, near, far, 1e-3, 0.004, rays_mask=mask, traverse_steps_limit=4,
                                n_alive=n_alive, t_sorted=ts, t_indices=ti, hits=hits, return_terminate=True)


for _ in range(20):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    call()
torch.cuda.synchronize()
print("per call", (time.perf_counter() - t0) / 200 * 1e6, "us")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    call()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
