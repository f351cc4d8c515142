#!/usr/bin/env python3
"""hipGraph capture of BASELINE cfg 3's fixed-shape step (PropNetEstimator.sampling level loop + proposal-loss backward):
eager vs replayed timings and equality of the results."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import nerfacc_amd as na

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
p = torch.nn.Parameter(torch.tensor([3.0, 4.0], device=dev))
est = na.PropNetEstimator().to(dev)
prop = lambda ts, te: torch.exp(-((ts + te) * 0.5 - p[1]) ** 2) * p[0]
fine = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0) ** 2 * 2.0) * 5.0


def step():
    ts, te = est.sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False, requires_grad=True)
    trans, _ = na.render_transmittance_from_density(ts, te, fine(ts, te))
    loss = est.compute_loss(trans)
    grad = torch.autograd.grad(loss, [p])[0]
    return ts, te, loss, grad


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# capture (before any eager run on the default stream)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ts, te, loss, grad = step()
g.replay()
torch.cuda.synchronize()
grads = [grad]
ts_g, te_g, loss_g, grad_g = ts.clone(), te.clone(), loss.clone(), grad.clone()
ts0, te0, loss0, g0 = step()
eager_ms = timeit(step)
ts, te, loss, grads = ts_g, te_g, loss_g, [grad_g]
same = bool(torch.equal(ts, ts0) and torch.equal(te, te0) and torch.equal(loss, loss0) and torch.equal(grads[0], g0))
graph_ms = timeit(g.replay)
print(json.dumps({"rays": R, "eager_ms": eager_ms, "graph_ms": graph_ms, "equal": same, "loss": float(loss), "grad": grads[0].tolist()}))
