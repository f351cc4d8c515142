"""Where does a limit-1 traversal on cfg 5's scene spend its time?  Sweep ray count, random density, binning."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerfacc_amd as na
from nerfacc_amd import grid as GR

dev = torch.device("cuda:0")
res, G = 512, 4
ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
aabbs = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=G).to(dev).aabbs


def scene(p):
    g = torch.Generator(device=dev); g.manual_seed(5)
    return torch.stack([((r > 0.5) & (r < 0.66)) | (torch.rand((res,) * 3, device=dev, generator=g) < p) for _ in range(G)])


def rays(R):
    rng = np.random.default_rng(5)
    o = torch.from_numpy(rng.random((R, 3)).astype(np.float32) - 0.5).to(dev)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return o, torch.from_numpy(d).to(dev)


def run(binaries, R, limit, bin_rays):
    o, d = rays(R)
    near = torch.full((R,), 0.2, device=dev); far = torch.full((R,), 1e10, device=dev)
    mask = torch.ones(R, dtype=torch.bool, device=dev)
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = GR._traverse_samples(o, d, binaries, aabbs, near, far, 1e-3, 0.004, rays_mask=mask, traverse_steps_limit=limit,
                                   n_alive=R, bin_rays=bin_rays)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return out[0].numel(), best


b02 = scene(0.02)
for R in (1 << 21, 1 << 20, 1 << 18, 1 << 16):
    for br in (None, False):
        n, ms = run(b02, R, 1, br)
        print(f"p=0.02 R={R:8d} bin={br!s:5s} limit 1: samples {n:9d} {ms:7.3f} ms")
for p in (0.0, 0.005, 0.1, 0.5):
    b = scene(p)
    n, ms = run(b, 1 << 21, 1, False)
    print(f"p={p:5.3f} R=2097152 bin=False limit 1: samples {n:9d} {ms:7.3f} ms")
