"""Where a limited cone walk spends its cycles (needs a build with -DNFA_CONE_PROFILE): set-up passes vs cell loop, lane use."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerfacc_amd as na
from nerfacc_amd import grid as GR

dev = torch.device("cuda:0")
res, G, R = 512, 4, 1 << 21
est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=G).to(dev)
ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
g = torch.Generator(device=dev); g.manual_seed(5)
est.binaries = torch.stack([((r > 0.5) & (r < 0.66)) | (torch.rand((res,) * 3, device=dev, generator=g) < 0.02) for _ in range(G)])
rng = np.random.default_rng(5)
o = torch.from_numpy(rng.random((R, 3)).astype(np.float32) - 0.5).to(dev)
d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
d = torch.from_numpy(d).to(dev)
near = torch.full((R,), 0.2, device=dev); far = torch.full((R,), 1e10, device=dev)
mask = torch.ones(R, dtype=torch.bool, device=dev)
tmin, tmax, hits = na.ray_aabb_intersect(o, d, est.aabbs)
ts, ti = torch.sort(torch.cat([tmin, tmax], -1), -1)
prof = torch.zeros(128 * 8, dtype=torch.int64, device=dev)
os.environ["NFA_CONE_PROFILE_PTR"] = str(prof.data_ptr())
for limit in (1, 4, 16):
    for rep in range(2):
        prof.zero_()
        torch.cuda.synchronize()
        out = GR._traverse_samples(o, d, est.binaries, est.aabbs, near, far, 1e-3, 0.004, rays_mask=mask, traverse_steps_limit=limit,
                                   n_alive=R, t_sorted=ts, t_indices=ti, hits=hits)
        torch.cuda.synchronize()
    p = prof.view(128, 8).sum(0).tolist()
    setup, cells, trips, lanes, rounds, waves = p[:6]
    print(f"limit {limit}: waves {waves}; per wave: set-up {setup / waves / 100:.0f} us-ish ({setup / waves:.0f} ticks), cells {cells / waves:.0f} ticks, "
          f"trips {trips / waves:.0f}, lanes per trip {lanes / max(trips, 1):.1f}, rounds {rounds / waves:.1f}; cell share {cells / (cells + setup):.2f}")
