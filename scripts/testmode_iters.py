"""Per-iteration cost of the test-mode loop on cfg 5's scene: alive rays, step limit, samples, traversal time."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as na
from nerfacc_amd import marching as M

dev = torch.device("cuda:0")
res, G, R = 512, 4, 1 << 21
est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=G).to(dev)
ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
g = torch.Generator(device=dev); g.manual_seed(5)
est.binaries = torch.stack([((r > 0.5) & (r < 0.66)) | (torch.rand((res,) * 3, device=dev, generator=g) < 0.02) for _ in range(G)])
est.occs = est.binaries.reshape(-1).float()
rng = np.random.default_rng(5)
o = torch.from_numpy(rng.random((R, 3)).astype(np.float32) - 0.5).to(dev)
d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
d = torch.from_numpy(d).to(dev)
fld = bench.NativeField(torch.nn.Parameter(torch.tensor([1.0, 1.0], device=dev)))
orig = M._traverse_samples
rows = []


def timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(*a, **k)
    e1.record(); torch.cuda.synchronize()
    rows.append((k.get("n_alive"), k.get("traverse_steps_limit"), out[0].numel(), e0.elapsed_time(e1)))
    return out


kw = dict(near_plane=0.2, render_step_size=1e-3, cone_angle=0.004, alpha_thre=1e-2, early_stop_eps=1e-4)
with torch.no_grad():
    M.render_rays_test_mode(1024, fld.rgb_sigma_fn, est, o, d, **kw)
    M._traverse_samples = timed
    M.render_rays_test_mode(1024, fld.rgb_sigma_fn, est, o, d, **kw)
print("iter  alive     limit  samples   traversal_ms")
for i, (a, l, n, ms) in enumerate(rows):
    if i < 12 or i % 8 == 0 or i == len(rows) - 1:
        print(f"{i:4d} {a:9d} {l:5d} {n:9d} {ms:8.3f}")
print("iterations", len(rows), "traversal total ms", sum(r[3] for r in rows))
