"""How the fused rendering passes depend on the ray length (same number of samples, uniform or mixed lengths)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerfacc_amd as na
from nerfacc_amd import _backend as B
from nerfacc_amd._segments import seginfo_from_packed

dev = torch.device("cuda:0")
M = 32 * 1024 * 1024
rng = np.random.default_rng(0)

def run(cnt, tag):
    cnt = cnt.astype(np.int64)
    n = int(cnt.sum()); R = cnt.size
    pi = torch.from_numpy(np.stack([np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt], -1)).to(dev)
    seg = seginfo_from_packed(pi, n)
    ts = torch.rand(n, device=dev); te = ts + 0.01
    sg = torch.rand(n, device=dev); rgb = torch.rand(n, 3, device=dev)
    w = torch.empty(n, device=dev); tr = torch.empty(n, device=dev); al = torch.empty(n, device=dev)
    col = torch.empty(R, 3, device=dev); op = torch.empty(R, device=dev); dp = torch.empty(R, device=dev)
    gs = torch.empty(n, device=dev); gr = torch.empty(n, 3, device=dev)
    gc = torch.rand(R, 3, device=dev)
    def fwd():
        B.call("nfa_render_fused_fwd", B.ptr(ts), B.ptr(te), B.ptr(sg), B.ptr(rgb), B.ptr(seg.packed_info), B.ptr(seg.tiles),
               seg.n_tiles, R, n, B.ptr(w), B.ptr(tr), B.ptr(al), B.ptr(col), B.ptr(op), B.ptr(dp), B.stream())
    def bwd():
        B.call("nfa_render_fused_bwd", B.ptr(ts), B.ptr(te), B.ptr(rgb), B.ptr(tr), B.ptr(al), B.ptr(gc), None, None, None, None, None,
               B.ptr(seg.packed_info), B.ptr(seg.tiles), seg.n_tiles, R, n, B.ptr(gs), B.ptr(gr), B.stream())
    vis_m = torch.empty(n, dtype=torch.uint8, device=dev); vis_c = torch.empty(R, dtype=torch.int64, device=dev)
    def vis():
        B.call("nfa_render_visibility", B.ptr(ts), B.ptr(te), B.ptr(sg), None, 1e-4, 0.0, B.ptr(seg.packed_info), B.ptr(seg.tiles),
               seg.n_tiles, R, n, B.ptr(vis_m), B.ptr(vis_c), B.stream())
    out = []
    for f in (fwd, bwd, vis):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{tag:34s} rays {R:9d} samples {n:10d}  fwd {out[0]:7.1f} us  bwd {out[1]:7.1f} us  vis {out[2]:7.1f} us", flush=True)

import bench
for res in (128, 256):
    w = bench.make_workload(dev, res=res)
    bench.run_step(w)
    cnt = torch.bincount(w["last"][0], minlength=w["n_rays"]).cpu().numpy()
    del w
    torch.cuda.empty_cache()
    run(cnt, f"bench rays, {res}^3 grid")
    run(cnt[cnt > 0], "  without the empty rays")
for L in (4, 16, 31, 32, 64, 1024):
    run(np.full(M // L, L), f"uniform {L}")
