# scratch: wall time of the test-mode marching loop (a12) on the bench workload
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import nerfacc_amd as na
from nerfacc_amd.marching import render_rays_test_mode
dev = torch.device("cuda:0")
w = bench.make_workload(dev)
fld = lambda ts, te, ri: w["rgb_sigma_fn"](ts, te, ri)
def run():
    with torch.no_grad():
        return render_rays_test_mode(1024, lambda ts, te, ri: tuple(x.detach() for x in fld(ts, te, ri)), w["estimator"],
                                     w["rays_o"], w["rays_d"], render_step_size=w["step"], early_stop_eps=1e-4)
for _ in range(2): out = run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): out = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("test-mode loop: %.2f ms per 1M-ray image, %d samples, %.1f M rays/s" % (dt * 1e3, out[3], w["n_rays"] / dt / 1e6))
# per-launch breakdown of one image
timer = bench.KernelTimer(); timer.install()
torch.cuda.synchronize()
out = run()
torch.cuda.synchronize()
ks = timer.summary(1); timer.uninstall()
tot = 0.0
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("  %-40s %7.2f ms  %4d launches" % (k, v["ms_per_step"], v["launches_per_step"]))
    tot += v["ms_per_step"]
print("  native total %.2f ms" % tot)
