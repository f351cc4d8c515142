"""Distribution of run records per ray (and per wave of 64 rays) on the bench workloads: sizes the one-pass kernel's LDS slots."""
import ctypes as C
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from nerfacc_amd import _backend as B
from nerfacc_amd import grid as G

dev = torch.device("cuda:0")
out = {}
for res, rays in ((128, "image"), (256, "image"), (128, "random")):
    w = bench.make_workload(dev, 1 << 20, res, "shell10", rays)
    est = w["estimator"]
    n = 1 << 20
    near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
    a = G._traverse_args(w["rays_o"], w["rays_d"], None, est.binaries, est.aabbs, None, None, None, near, far, w["step"], 0.0, -1, 0)
    sm = torch.empty(n, dtype=torch.int64, device=dev)
    a.sm_cnts = B.ptr(sm)
    bits = G._get_walk_bits(est.binaries)
    rc = torch.empty(n, dtype=torch.int32, device=dev)
    runs = torch.empty((32, n), dtype=torch.int64, device=dev)
    ov = torch.zeros(1, dtype=torch.int32, device=dev)
    B.call("nfa_traverse_runs", C.byref(a), B.ptr(bits), B.ptr(rc), B.ptr(runs), 32, B.ptr(ov), 0.0, None, 0, B.stream())
    torch.cuda.synchronize()
    c = rc.cpu().numpy()
    s = sm.cpu().numpy()
    hist = np.bincount(np.minimum(c, 40), minlength=41)
    wave_max = c.reshape(-1, 64).max(1)
    wave_sum = c.reshape(-1, 64).sum(1)
    out[f"{res}_{rays}"] = dict(mean=float(c.mean()), p50=int(np.percentile(c, 50)), p90=int(np.percentile(c, 90)), p99=int(np.percentile(c, 99)),
                                max=int(c.max()), frac_gt8=float((c > 8).mean()), frac_gt6=float((c > 6).mean()), frac_gt12=float((c > 12).mean()),
                                hist=hist.tolist(), wave_sum_mean=float(wave_sum.mean()), wave_sum_p99=int(np.percentile(wave_sum, 99)),
                                wave_max_mean=float(wave_max.mean()), samples=int(s.sum()),
                                wave_samples_mean=float(s.reshape(-1, 64).sum(1).mean()), wave_samples_max=int(s.reshape(-1, 64).sum(1).max()))
    np.save(f"gpurun_out/r03_cnt_{res}_{rays}.npy", s.astype(np.int32))
print(json.dumps(out))
