"""scratch: how much of the run-length walk's cost on unrelated rays is lane incoherence? (rays pre-sorted by keys)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import nerfacc_amd as na
dev = torch.device("cuda:0")
w = bench.make_workload(dev, 1 << 20, 128, "shell10", "random", 0, "native")
o, d = w["rays_o"], w["rays_d"]
def walk_ms(o, d):
    est = w["estimator"]
    f = lambda: na.grid._traverse_samples(o, d, est.binaries, est.aabbs, torch.zeros(o.shape[0], device=dev),
                                          torch.full((o.shape[0],), 1e10, device=dev), w["step"], 0.0, near_hint=0.0)
    for _ in range(3): f()
    t = bench.KernelTimer(); t.install(); torch.cuda.synchronize()
    for _ in range(10): out = f()
    torch.cuda.synchronize(); ks = t.summary(10); t.uninstall()
    return {k: round(v["ms_per_step"] * 1e3) for k, v in ks.items()}, out[0].numel()
print("unsorted", walk_ms(o, d))
# keys: entry cell (coarse) + direction octant
tmin, tmax, hit = na.ray_aabb_intersect(o, d, w["estimator"].aabbs)
tin = torch.clamp(tmin[:, 0], min=0.0)
pin = o + d * tin[:, None]
for bits in (2, 3, 4):
    q = ((pin.clamp(-1, 1) + 1) * 0.5 * (1 << bits)).long().clamp(0, (1 << bits) - 1)
    oct_ = ((d[:, 0] > 0).long() << 2) | ((d[:, 1] > 0).long() << 1) | (d[:, 2] > 0).long()
    key = (((q[:, 0] << bits | q[:, 1]) << bits | q[:, 2]) << 3 | oct_)
    key = torch.where(hit[:, 0], key, torch.full_like(key, 1 << 20))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    order = torch.sort(key.int(), stable=True)[1]
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) * 1e3
    print("bits", bits, "sort %.3f ms" % ts, walk_ms(o[order].contiguous(), d[order].contiguous()))
# path length as key
plen = torch.where(hit[:, 0], (tmax[:, 0] - tin), torch.zeros_like(tin))
order = torch.sort(plen)[1]
print("by path length", walk_ms(o[order].contiguous(), d[order].contiguous()))
