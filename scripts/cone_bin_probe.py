"""Cone-angle walk: does binning by crossed cells hurt image-ordered (coherent) rays?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as na
from nerfacc_amd import grid as G

dev = torch.device("cuda:0")
res, levels = 256, 3
est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
g = torch.Generator(device=dev); g.manual_seed(5)
est.binaries = torch.stack([((r > 0.5) & (r < 0.66)) | (torch.rand((res,) * 3, device=dev, generator=g) < 0.02) for _ in range(levels)])
n = 1 << 20
for name, (o, d) in (("image, camera outside", bench.make_rays(n, "image")),
                     ("image, camera inside", (np.zeros((n, 3), np.float32) + np.float32(0.1), bench.make_rays(n, "image")[1])),
                     ("random from inside", ((np.random.default_rng(1).random((n, 3)).astype(np.float32) - 0.5), bench.make_rays(n, "random")[1]))):
    ro, rd = torch.from_numpy(np.ascontiguousarray(o)).to(dev), torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    near, far = torch.full((n,), 0.05, device=dev), torch.full((n,), 1e10, device=dev)
    outs = []
    for binned in (False, True):
        ts = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = G._traverse_samples(ro, rd, est.binaries, est.aabbs, near, far, 2e-3, 0.004, near_hint=0.05, bin_rays=binned)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        outs.append(out)
        print(f"{name:24s} binned={binned!s:5s} traversal ms: {min(ts):7.3f}   samples {out[0].numel()}")
    assert all(torch.equal(a, b) for a, b in zip(*outs))
