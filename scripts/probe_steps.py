"""scratch: per-step wall times of the sequential and the pipelined bench loops (finds one-off stalls)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
w = bench.make_workload(dev, 1024 * 1024, 128, "shell10", "image", 0, "native")
def timed(fn, n):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return out
print("seq ", " ".join("%.2f" % t for t in timed(lambda: bench.run_step(w, 1), 30)))
h = [w["estimator"].prefetch_traversal(w["rays_o"], w["rays_d"], render_step_size=w["step"], wait_for_inputs=False)]
def pstep():
    _, _, h[0] = bench.run_step(w, 1, h[0], prefetch=True)
print("pipe", " ".join("%.2f" % t for t in timed(pstep, 30)))
import gc
print("gc objects", len(gc.get_objects()), gc.get_count(), gc.get_threshold())
mode = os.environ.get("PROBE_GC", "")
if mode == "freeze":
    gc.collect(); gc.freeze()
elif mode == "disable":
    gc.disable()
gc.callbacks.append(lambda phase, info: phase == "stop" and info["generation"] == 2 and print("  [gc gen2]", info, flush=True))
# unsynchronised blocks of 5
for name, fn in (("seq", lambda: bench.run_step(w, 1)), ("pipe", pstep)):
    res = []
    for _ in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 5 * 1e3)
    print(name + "5", " ".join("%.2f" % t for t in res))
