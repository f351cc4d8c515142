"""Scratch: run the sampler's traversal a few times (for rocprofv3 --pmc)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, nerfacc_amd as na
dev = torch.device("cuda:0")
R = 1024 * 1024
w = bench.make_workload(dev, R, 128, grid=(sys.argv[1] if len(sys.argv) > 1 else "shell10"))
near = torch.zeros(R, device=dev); far = torch.full((R,), 1e10, device=dev)
for _ in range(4):
    out = na.grid._traverse_samples(w["rays_o"], w["rays_d"], w["estimator"].binaries, w["estimator"].aabbs, near, far, w["step"], 0.0)
torch.cuda.synchronize()
print("M", out[0].numel())
