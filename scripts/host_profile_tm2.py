"""Host-side cost of one test-mode iteration on a small constant-step scene (python + launches, GPU work negligible)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfacc_amd.marching import render_rays_test_mode

dev = torch.device("cuda:0")
w = bench.make_workload(dev, 256 * 256, 128, "shell10", "image", 0, "native")
est = w["estimator"]
kw = dict(render_step_size=w["step"], early_stop_eps=1e-4)
with torch.no_grad():
    for _ in range(3):
        out = render_rays_test_mode(1024, w["rgb_sigma_fn"], est, w["rays_o"], w["rays_d"], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = render_rays_test_mode(1024, w["rgb_sigma_fn"], est, w["rays_o"], w["rays_d"], **kw)
    torch.cuda.synchronize()
    print("per image", (time.perf_counter() - t0) / 10 * 1e3, "ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        out = render_rays_test_mode(1024, w["rgb_sigma_fn"], est, w["rays_o"], w["rays_d"], **kw)
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(32)
