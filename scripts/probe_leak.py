# scratch: device memory must not grow across steps (sequential and pipelined)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
w = bench.make_workload(dev)
def mem(): torch.cuda.synchronize(); return torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20
for _ in range(5): bench.run_step(w)
print("after warmup      alloc %.0f MiB reserved %.0f MiB" % mem())
for _ in range(150): bench.run_step(w)
print("after 150 steps   alloc %.0f MiB reserved %.0f MiB" % mem())
h = w["estimator"].prefetch_traversal(w["rays_o"], w["rays_d"], render_step_size=w["step"], wait_for_inputs=False)
for _ in range(150): _, _, h = bench.run_step(w, 1, h, prefetch=True)
del h
print("after 150 pipelined alloc %.0f MiB reserved %.0f MiB" % mem())
