"""Where the host spends its time in one headline step (cProfile over 60 steps, GPU work included as sync time)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
w = bench.make_workload(dev)
for _ in range(5):
    bench.run_step(w)
torch.cuda.synchronize()
import gc
gc.disable()
t0 = time.perf_counter()
for _ in range(60):
    bench.run_step(w)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 60 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(60):
    bench.run_step(w)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
