"""Cost of one limited traversal on cfg 5's scene as a function of the step limit (all rays alive, from near = 0.2)."""
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
sink = {}
LIMITS = tuple(int(v) for v in sys.argv[1:]) or (1, 4, 16, 64, 256, 0)
for limit in LIMITS:
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = GR._traverse_samples(o, d, est.binaries, est.aabbs, near, far, 1e-3, 0.004, rays_mask=mask if limit else None,
                                   traverse_steps_limit=limit, n_alive=R if limit else None,
                                   bin_rays={"": None, "0": False, "1": True}[os.environ.get("NFA_LS_BIN", "")], stats_sink=sink)
        e1.record(); torch.cuda.synchronize()
    print(f"limit {limit:4d}: samples {out[0].numel():10d}  {e0.elapsed_time(e1):8.3f} ms")
