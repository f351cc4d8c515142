"""scratch: the sampler's traversal with stratified=True (per-ray near planes: no approach table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import nerfacc_amd as na
dev = torch.device("cuda:0")
w = bench.make_workload(dev)
est = w["estimator"]
for strat in (False, True):
    f = lambda: est.sampling(w["rays_o"], w["rays_d"], render_step_size=w["step"], stratified=strat)
    for _ in range(3): f()
    t = bench.KernelTimer(); t.install(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ks = t.summary(10); t.uninstall()
    print("stratified", strat, "%.3f ms" % (dt * 1e3), out[0].numel(), {k: round(v["ms_per_step"] * 1e3) for k, v in ks.items()})
