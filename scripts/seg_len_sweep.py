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
    out = []
    for f in (fwd, bwd):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{tag:34s} rays {R:9d} samples {n:10d}  fwd {out[0]:7.1f} us  bwd {out[1]:7.1f} us   per Msample {out[0]/n*1e6:6.2f} {out[1]/n*1e6:6.2f}", flush=True)

import bench
w = bench.make_workload(dev, res=256)
bench.run_step(w)
cnt = torch.bincount(w["last"][0], minlength=w["n_rays"]).cpu().numpy()
del w
torch.cuda.empty_cache()
run(cnt, "bench rays, 256^3 grid")
ne = cnt[cnt > 0]; nz = int((cnt == 0).sum())
run(ne, "  without the empty rays")
run(np.concatenate([ne, np.zeros(nz, np.int64)]), "  empties moved to the end")
run(np.concatenate([np.zeros(nz, np.int64), ne]), "  empties moved to the front")
# empties spread evenly: one after every k-th non-empty ray
k = max(1, ne.size // nz)
ev = np.insert(ne, np.arange(k, k * nz + 1, k)[:nz].clip(max=ne.size), 0)
run(ev, f"  empties spread evenly (every {k} rays)")
# runs of empties as in the bench but the non-empty rays replaced by uniform 38
u = cnt.copy(); u[u > 0] = 38
run(u, "  same empties, other rays uniform 38")
# run-length stats of the empties
z = (cnt == 0).astype(np.int8); d = np.diff(np.concatenate([[0], z, [0]])); st = np.where(d == 1)[0]; en = np.where(d == -1)[0]; rl = en - st
print("empty runs:", rl.size, "mean", rl.mean(), "pct", np.percentile(rl, [50, 90, 99, 100]).tolist(), flush=True)
