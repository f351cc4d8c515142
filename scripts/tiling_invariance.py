"""Results of the packed ops must not depend on the tiling: run with NFA_SEG_TILE=256 / 1024 / 4096 in the environment (handed to nfa_set_tuning)
and compare the digests.  Random ragged batches: empty rays, runs of tiny rays, long rays."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerfacc_amd as na
from nerfacc_amd import _backend as NB
NB.set_tuning("NFA_SEG_TILE", os.environ.get("NFA_SEG_TILE"))   # (the knob of include/nerfacc_hip.h: nfa_set_tuning)
dev = torch.device("cuda:0")
h = hashlib.sha256()
for seed in range(12):
    rng = np.random.default_rng(seed)
    R = int(rng.integers(1, 40000))
    kind = seed % 4
    if kind == 0: cnt = rng.poisson(rng.uniform(0.2, 40), R)
    elif kind == 1: cnt = rng.choice([0, 0, 0, 1, 2, 3, 200], R, p=[.3, .2, .1, .15, .1, .1, .05])
    elif kind == 2: cnt = np.where(rng.random(R) < 0.02, rng.integers(500, 5000, R), rng.integers(0, 4, R))
    else: cnt = rng.integers(0, 70, R)
    cnt = cnt.astype(np.int64)
    n = int(cnt.sum())
    if n == 0: continue
    ri = torch.from_numpy(np.repeat(np.arange(R), cnt)).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    ts = torch.rand(n, generator=g, device=dev); te = ts + 0.03
    sig = (torch.rand(n, generator=g, device=dev) * 5).requires_grad_(True)
    rgb = torch.rand(n, 3, generator=g, device=dev).requires_grad_(True)
    col, op, dp, ex = na.rendering(ts, te, ri, n_rays=R, rgb_sigma_fn=lambda a, b, c: (rgb, sig))
    gw = torch.rand(R, 3, generator=g, device=dev)
    ((col * gw).sum() + (dp * gw[:, :1]).sum() + op.sum()).backward()
    vis = na.render_visibility_from_density(ts, te, sig.detach(), ray_indices=ri, n_rays=R, early_stop_eps=1e-2)
    w2, tr2, al2 = na.render_weight_from_density(ts, te, sig.detach(), ray_indices=ri, n_rays=R)
    acc = na.accumulate_along_rays(w2, rgb.detach(), ray_indices=ri, n_rays=R)
    ex_s = na.exclusive_sum(sig.detach(), packed_info=na.pack_info(ri, R))
    for t in (col, op, dp, ex["weights"], ex["trans"], sig.grad, rgb.grad, vis, w2, tr2, acc, ex_s):
        h.update(t.detach().cpu().numpy().tobytes())
print("digest", h.hexdigest())
