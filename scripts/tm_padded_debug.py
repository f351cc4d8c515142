"""debugging aid: the padded test-mode loop's iteration call by call, with a synchronisation after each native call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerfacc_amd as na
from nerfacc_amd import _backend as B
from nerfacc_amd.marching import PaddedTestModeLoop
dev = torch.device("cuda:0")
rng = np.random.default_rng(23)
n, res, step = 6000, 48, 6e-3
o = (rng.random((n, 3)).astype(np.float32) - 0.5) * 3.0
d = rng.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
b = rng.random((1, res, res, res)) < 0.25
est = na.OccGridEstimator([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=1).to(dev)
est.binaries = torch.from_numpy(b).to(dev)
def field_t(ts, te, ri):
    tm = (ts + te) * 0.5
    return (torch.stack([0.5 + 0.5 * torch.cos(tm), (ri % 7).float() / 7.0, torch.full_like(tm, 0.3)], -1), 25.0 * (0.5 + 0.5 * torch.sin(9.0 * tm)))
loop = PaddedTestModeLoop(600, field_t, est, torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 0.05, 1e10, step, 0.0, 1e-3, use_graph=False)
orig = B.call
def call(name, *a):
    print("call", name, flush=True)
    orig(name, *a)
    torch.cuda.synchronize()
    print("  ok", flush=True)
B.call = call
import nerfacc_amd.marching as M
M.B.call = call
loop._reset(); torch.cuda.synchronize()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print("iteration", it, flush=True)
    loop._iteration()
    torch.cuda.synchronize()
    print("  state", loop.state.tolist(), "alive", int(loop.alive_count[0]), "meta", loop.meta.tolist(), flush=True)
if len(sys.argv) > 2:
    B.call = orig; M.B.call = orig
    loop2 = PaddedTestModeLoop(600, field_t, est, torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 0.05, 1e10, step, 0.0, 1e-3, use_graph=True)
    print("capturing", flush=True)
    loop2._reset()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        loop2._iteration()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loop2._iteration()
    print("captured", flush=True)
    loop2._reset(); torch.cuda.synchronize()
    for it in range(int(sys.argv[2])):
        g.replay(); torch.cuda.synchronize()
        print("replay", it, loop2.state.tolist(), int(loop2.alive_count[0]), flush=True)
    print("back to back", flush=True)
    loop2._reset(); torch.cuda.synchronize()
    for it in range(8):
        g.replay()
    torch.cuda.synchronize()
    print("8 replays ok", loop2.state.tolist(), flush=True)
    host = torch.empty(2, dtype=torch.int64, pin_memory=True)
    for it in range(4):
        g.replay()
    host[0:1].copy_(loop2.alive_count[0:1], non_blocking=True)
    ev = torch.cuda.Event(); ev.record()
    for it in range(4):
        g.replay()
    ev.synchronize()
    print("event ok", host.tolist(), flush=True)
    torch.cuda.synchronize()
    loop2.graph = g
    out = loop2.render(None)
    print("render ok", out[3], loop2.iterations_run, loop2.iterations_queued, flush=True)
