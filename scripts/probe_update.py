"""scratch: cost of OccGridEstimator.update_every_n_steps (grid maintenance, plain torch as upstream)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerfacc_amd as na
dev = torch.device("cuda:0")
for res, levels in ((128, 1), (128, 4)):
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=levels).to(dev)
    occ_fn = lambda x: (torch.exp(-((x.norm(dim=-1, keepdim=True) - 0.6) ** 2) * 50.0)) * 0.05
    for step in range(0, 16 * 20, 16):            # warm-up phase (all cells)
        est.update_every_n_steps(step, occ_fn, occ_thre=0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for step in range(1024, 1024 + 16 * 20, 16):  # sampling phase
        est.update_every_n_steps(step, occ_fn, occ_thre=0.01)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("res %d levels %d: %.3f ms per update (occupancy %.3f)" % (res, levels, dt * 1e3, est.binaries.float().mean().item()))
