#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json: cfg 3 (PropNetEstimator, 1 M rays, 64 -> 16 samples,
fwd + bwd of the proposal loss) and cfg 5 (4 nested 512^3 levels, 2 M rays from inside the level-0 box,
cone_angle 0.004, alpha_thre 1e-2: train-mode sampling + rendering fwd/bwd).  Not the headline bench (bench.py):
prints one JSON object per config with wall time per step and per-native-call HIP-event times.

    python scripts/bench_configs.py [cfg3] [cfg5] [--steps K] [--warmup W] [--rays R]
"""
import argparse
import gc
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nerfacc_amd as na  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    timer = bench.KernelTimer(); timer.install()
    gc.collect(); gc.disable()   # as in bench.py: no interpreter GC pause inside the timed region
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gc.enable()
    ks = timer.summary(steps); timer.uninstall()
    return dt, ks, out


def cfg3(dev, R, steps, warmup):
    p = torch.nn.Parameter(torch.tensor([3.0, 4.0], device=dev))
    est = na.PropNetEstimator(optimizer=torch.optim.SGD([p], lr=1e-3)).to(dev)
    prop = lambda ts, te: torch.exp(-((ts + te) * 0.5 - p[1]) ** 2) * p[0]          # proposal density, 2 parameters
    fine = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0) ** 2 * 2.0) * 5.0

    def step():
        ts, te = est.sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False,
                              requires_grad=True)
        trans, _ = na.render_transmittance_from_density(ts, te, fine(ts, te))
        return est.update_every_n_steps(trans, requires_grad=True)

    dt, ks, loss = timed(step, steps, warmup)
    return dict(config="cfg3: PropNetEstimator 2 proposal levels 64->64->16, R=%d, uniform, fwd + proposal-loss bwd" % R,
                ms_per_step=dt * 1e3, rays_per_s=R / dt, loss=float(loss), native=ks)


def cfg5(dev, R, steps, warmup, res=512, G=4):
    rng = np.random.default_rng(5)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=G).to(dev)
    # ~10 % shell per level (in level-local coordinates), built on the device level by level (4 x 128 MiB bool)
    ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
    shell = (r > 0.5) & (r < 0.66)
    g = torch.Generator(device=dev); g.manual_seed(5)
    b = torch.stack([shell | (torch.rand((res, res, res), device=dev, generator=g) < 0.02) for _ in range(G)])
    est.binaries = b
    est.occs = b.reshape(-1).float()
    del r, shell
    o = (rng.random((R, 3)).astype(np.float32) - 0.5)                               # inside the level-0 box
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rays_o, rays_d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    params = torch.nn.Parameter(torch.tensor([1.0, 1.0], device=dev))
    fld = bench.NativeField(params)

    def step():
        ri, ts, te = est.sampling(rays_o, rays_d, sigma_fn=fld.sigma_fn, near_plane=0.2, render_step_size=1e-3,
                                  cone_angle=0.004, alpha_thre=1e-2, early_stop_eps=1e-4)
        colors, opac, depth, _ = na.rendering(ts, te, ri, n_rays=R, rgb_sigma_fn=fld.rgb_sigma_fn)
        params.grad = None
        colors.sum().backward()
        return ri.numel()

    dt, ks, m = timed(step, steps, warmup)
    return dict(config="cfg5: %d nested %d^3 levels, R=%d rays from inside, step 1e-3, cone 0.004, near 0.2, alpha_thre 1e-2, "
                       "sampling + rendering fwd + bwd" % (G, res, R),
                ms_per_step=dt * 1e3, rays_per_s=R / dt, samples_after_compaction=int(m), native=ks)


def api_traverse(dev, R, steps, warmup):
    """The API-faithful traverse_grids (intervals + samples + masks) on the cfg-2 workload."""
    w = bench.make_workload(dev, R)
    est = w["estimator"]

    def step():
        iv, sm, _ = na.traverse_grids(w["rays_o"], w["rays_d"], est.binaries, est.aabbs, step_size=w["step"])
        return iv.vals.numel(), sm.vals.numel()

    dt, ks, (E, M) = timed(step, steps, warmup)
    nbytes = 14 * E + 13 * M + 36 * R + R * 32 + est.binaries.numel()
    return dict(config="api traverse_grids on cfg2: R=%d, E=%d edges, M=%d samples" % (R, E, M), ms_per_step=dt * 1e3,
                rays_per_s=R / dt, algorithmic_bytes=nbytes, achieved_GBps=nbytes / dt / 1e9, native=ks)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cfgs", nargs="*", default=["cfg3", "cfg5"])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=0)
    ap.add_argument("--res", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for c in a.cfgs:
        if c == "cfg3":
            out = cfg3(dev, a.rays or 1 << 20, a.steps, a.warmup)
        elif c == "trav":
            out = api_traverse(dev, a.rays or 1 << 20, a.steps, a.warmup)
        else:
            out = cfg5(dev, a.rays or 1 << 21, a.steps, a.warmup, res=a.res)
        print(json.dumps(out), flush=True)
