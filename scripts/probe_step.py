"""scratch: nfa_render_step_accumulate on controlled shapes (rays x samples per ray)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerfacc_amd as na
from nerfacc_amd._segments import tag_trusted
from nerfacc_amd.marching import _render_step_native
dev = torch.device("cuda:0")
R = 1 << 20
for per, frac in ((1, 1.0), (2, 1.0), (4, 1.0), (32, 1.0), (1, 0.25), (64, 0.02)):
    cnts = torch.zeros(R, dtype=torch.int64, device=dev)
    alive = torch.rand(R, device=dev) < frac
    cnts[alive] = per
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    packed = na.grid._cumsum_packed(cnts, total)
    n = int(total)
    tag_trusted(packed, n)
    seg = packed._nfa_seg[2]
    ts = torch.rand(n, device=dev); te = ts + 0.01
    sig = torch.rand(n, device=dev) * 3; rgbs = torch.rand(n, 3, device=dev)
    rgb = torch.zeros(R, 3, device=dev); op = torch.zeros(R, 1, device=dev); dp = torch.zeros(R, 1, device=dev)
    for _ in range(3): _render_step_native(seg, ts, te, sig, rgbs, 0.0, rgb, op, dp, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): _render_step_native(seg, ts, te, sig, rgbs, 0.0, rgb, op, dp, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("rays %d x %d samples (%.0f %% alive): n=%d  %.3f ms  (%.0f GB/s on 24 B/sample + 36 B/ray)" % (R, per, frac * 100, n, dt * 1e3, (24 * n + 36 * R) / dt / 1e9))
