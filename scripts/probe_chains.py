"""scratch: distribution of continuous chains (runs) per ray on the cfg-5 geometry with a cone angle"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerfacc_amd as na
dev = torch.device("cuda", 0)
res, G, R = 512, 4, 1 << 17
rng = np.random.default_rng(5)
est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=G).to(dev)
ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
shell = (r > 0.5) & (r < 0.66)
g = torch.Generator(device=dev); g.manual_seed(5)
for speckle in (0.02, 0.0):
    b = torch.stack([shell | (torch.rand((res, res, res), device=dev, generator=g) < speckle) for _ in range(G)])
    o = (rng.random((R, 3)).astype(np.float32) - 0.5)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    iv, sm, _ = na.traverse_grids(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), b, est.aabbs,
                                  near_planes=torch.full((R,), 0.2, device=dev), step_size=1e-3, cone_angle=0.004)
    chains = (iv.packed_info[:, 1] - sm.packed_info[:, 1]).float()
    cnt = sm.packed_info[:, 1].float()
    sub = chains + cnt / 64
    q = torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)
    print("speckle", speckle, "samples/ray mean %.0f" % cnt.mean().item(), "chains/ray quantiles", torch.quantile(chains, q).tolist(),
          "records (chains + samples/64) quantiles", torch.quantile(sub, q).tolist(), "frac > 32: %.3f  > 64: %.3f" % ((sub > 32).float().mean().item(), (sub > 64).float().mean().item()))
