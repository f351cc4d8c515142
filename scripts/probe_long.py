"""scratch: engine ops on very long rays (a ray is owned by one wave: no cross-workgroup carry)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerfacc_amd as na
dev = torch.device("cuda:0")
for R, per in ((1 << 20, 32), (1 << 15, 1024), (1 << 12, 8192), (512, 65536), (32, 1 << 20)):
    n = R * per
    ri = torch.arange(R, device=dev).repeat_interleave(per)
    ts = torch.rand(n, device=dev); te = ts + 0.01; sig = torch.rand(n, device=dev) * 0.01
    f = lambda: na.render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=R)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%8d rays x %8d samples: %.3f ms (%.0f GB/s)" % (R, per, dt * 1e3, 24 * n / dt / 1e9))
