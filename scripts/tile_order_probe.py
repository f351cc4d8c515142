"""Does the walk gain from waves of 8x8-pixel tiles instead of 64x1 row segments?  (lane -> ray assignment through the walk's
ray_order argument; results do not depend on it)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfacc_amd import _backend as B
from nerfacc_amd import grid as G

dev = torch.device("cuda:0")
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = bench.make_workload(dev, 1 << 20, res, "shell10", "image")
est = w["estimator"]
n = 1 << 20
side = 1024
near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
bits = G._get_walk_bits(est.binaries)


def order_for(tw, th, wg_w=None):
    """rays of a wave = a tw x th pixel tile (tw * th == 64); the 4 waves of a workgroup = horizontally adjacent tiles"""
    ys, xs = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    ty, tx = ys // th, xs // tw
    iy, ix = ys % th, xs % tw
    key = ((ty * (side // tw) + tx) * th + iy) * tw + ix
    order = np.empty(n, np.int32)
    order[key.reshape(-1)] = np.arange(n, dtype=np.int32)
    return torch.from_numpy(order).to(dev)


ref = None
for name, order in (("64x1 (row-major)", None), ("32x2", order_for(32, 2)), ("16x4", order_for(16, 4)), ("8x8", order_for(8, 8)), ("4x16", order_for(4, 16))):
    a = G._traverse_args(w["rays_o"], w["rays_d"], None, est.binaries, est.aabbs, None, None, None, near, far, w["step"], 0.0, -1, 0)
    sm = torch.empty(n, dtype=torch.int64, device=dev)
    a.sm_cnts = B.ptr(sm)
    rc = torch.empty(n, dtype=torch.int32, device=dev)
    runs = torch.empty((32, n), dtype=torch.int64, device=dev)
    ov = torch.zeros(1, dtype=torch.int32, device=dev)
    ts = []
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        B.call("nfa_traverse_runs", C.byref(a), B.ptr(bits), B.ptr(rc), B.ptr(runs), 32, B.ptr(ov), 0.0, B.ptr(order), 0 if order is None else n, B.stream())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    if ref is None:
        ref = sm.clone()
    assert torch.equal(ref, sm)
    print(f"{name:18s} walk us: min {min(ts):7.1f} median {sorted(ts)[2]:7.1f}")
