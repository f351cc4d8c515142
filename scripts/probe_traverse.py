"""Scratch probe: time the runs pass / expansion under controlled conditions."""
import ctypes as C, math, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nerfacc_amd as na
from nerfacc_amd import _backend as B, grid as G

dev = torch.device("cuda:0")
R = 1024 * 1024
step = 2 * math.sqrt(3) / 1024

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def runs_only(o, d, b, near, far, label):
    ab = torch.tensor([[-1., -1, -1, 1, 1, 1]], device=dev)
    (dev_, o, d, b, ab, near, far, _, ts, ti, hits) = G._prepare(o, d, b, ab, near, far, None, None, None, None, True)
    bricks, coarse = G._get_bricks(b)
    sm_cnts = torch.empty(R, dtype=torch.int64, device=dev)
    run_cnts = torch.empty(R, dtype=torch.int32, device=dev)
    runs = torch.empty((R, 32), dtype=torch.int64, device=dev)
    meta = torch.zeros(2, dtype=torch.int64, device=dev)
    a = G._traverse_args(o, d, None, b, ab, ts, ti, hits, near, far, step, 0.0, -1, 0)
    a.sm_cnts = B.ptr(sm_cnts)
    def f():
        B.call("nfa_traverse_runs", C.byref(a), B.ptr(bricks), B.ptr(coarse), B.ptr(run_cnts), B.ptr(runs), 32, B.ptr(meta[1:2]), B.stream())
    t = timeit(f)
    def g():
        a.mode = 0
        G._launch(a)
    t_old = timeit(g)
    print(f"{label:40s} runs pass {t:8.1f} us   v1 count pass {t_old:8.1f} us   M={int(sm_cnts.sum())}  runs/ray={float(run_cnts.float().mean()):.2f} max={int(run_cnts.max())}")

w = bench.make_workload(dev, R, 128)
o, d = w["rays_o"], w["rays_d"]
near0 = torch.zeros(R, device=dev); far = torch.full((R,), 1e10, device=dev)
res = 128
c = (np.arange(res) + 0.5) / res * 2 - 1
x, y, z = np.meshgrid(c, c, c, indexing="ij")
r = np.sqrt(x * x + y * y + z * z)
rng = np.random.default_rng(0)
grids = {
  "shell only": ((r >= 0.50) & (r <= 0.66))[None],
  "speckle 2% only": (rng.random((1, res, res, res)) < 0.02),
  "speckle 0.2% only": (rng.random((1, res, res, res)) < 0.002),
  "full ones": np.ones((1, res, res, res), bool),
  "slab z in [0,0.1]": ((z >= 0) & (z <= 0.1))[None],
  "slab z in [0,0.5]": ((z >= 0) & (z <= 0.5))[None],
}
for k, g in list(grids.items())[:3]:
    runs_only(o, d, torch.from_numpy(g).to(dev), near0, far, k)
