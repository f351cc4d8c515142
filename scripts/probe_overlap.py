# scratch: does the VALU-bound walk overlap with the HBM-bound rendering passes when issued on two streams?
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import nerfacc_amd as na

dev = torch.device("cuda:0")
w = bench.make_workload(dev)
est, n = w["estimator"], w["n_rays"]
near = torch.zeros(n, device=dev); far = torch.full((n,), 1e10, device=dev)
ri, ts, te = est.sampling(w["rays_o"], w["rays_d"], sigma_fn=w["sigma_fn"], render_step_size=w["step"], early_stop_eps=1e-4)
side = torch.cuda.Stream()

def render():
    colors, _, _, _ = na.rendering(ts, te, ri, n_rays=n, rgb_sigma_fn=w["rgb_sigma_fn"])
    w["params"].grad = None
    colors.sum().backward()

def traverse():
    return na.grid._traverse_samples(w["rays_o"], w["rays_d"], est.binaries, est.aabbs, near, far, w["step"], 0.0, near_hint=0.0)

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

t_r, t_t = timeit(render), timeit(traverse)
def both_seq():
    render(); traverse()
def both_par():
    render()                       # queued on the main stream, returns at once
    with torch.cuda.stream(side):  # the walk + expansion on a second stream; the host read inside waits for it
        traverse()
t_s, t_p = timeit(both_seq), timeit(both_par)
print("render %.3f ms  traverse %.3f ms  sequential %.3f ms  two streams %.3f ms" % (t_r, t_t, t_s, t_p))
