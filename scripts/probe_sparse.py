"""scratch: engine ops on packed_info with long runs of empty rays (image-order batches with background)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerfacc_amd as na
dev = torch.device("cuda:0")
R = 1 << 20
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, cnts in (
    ("dense: every ray 8 samples", torch.full((R,), 8, dtype=torch.int64)),
    ("one block alive: rays [400k,600k) x 42", torch.cat([torch.zeros(400_000, dtype=torch.int64), torch.full((200_000,), 42, dtype=torch.int64), torch.zeros(R - 600_000, dtype=torch.int64)])),
    ("rows of 1024: 300 background + 724 x 12", torch.tensor(([0] * 300 + [12] * 724) * (R // 1024), dtype=torch.int64)),
    ("random 50 % empty", (torch.rand(R) < 0.5).long() * 16),
):
    cnts = cnts.to(dev)
    ri = torch.repeat_interleave(torch.arange(R, device=dev), cnts)
    n = ri.numel()
    ts = torch.rand(n, device=dev); te = ts + 0.01; sig = torch.rand(n, device=dev) * 3; rgbs = torch.rand(n, 3, device=dev)
    t_w = bench(lambda: na.render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=R))
    w = na.render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=R)[0]
    t_a = bench(lambda: na.accumulate_along_rays(w, rgbs, ri, R))
    t_r = bench(lambda: na.rendering(ts, te, ri, n_rays=R, rgb_sigma_fn=lambda a, b, c: (rgbs, sig)))
    print("%-44s n=%9d  weights %.3f ms  accumulate %.3f ms  rendering %.3f ms" % (name, n, t_w, t_a, t_r))
