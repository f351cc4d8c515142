"""Timing probe: the constant-step traversal in C chunks of rays, chunk k's cumsum + expansion on a second stream under chunk
k + 1's walk (each chunk into arrays of its own: timing only)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerfacc_amd import _backend as B
from nerfacc_amd import grid as G

dev = torch.device("cuda:0")
w = bench.make_workload(dev, 1 << 20, 128, "shell10", "image")
est = w["estimator"]
n = 1 << 20
bits = G._get_walk_bits(est.binaries)
near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def run(chunks, overlap):
    per = n // chunks
    bufs = []
    for c in range(chunks):
        sl = slice(c * per, (c + 1) * per)
        a = G._traverse_args(w["rays_o"][sl], w["rays_d"][sl], None, est.binaries, est.aabbs, None, None, None, near[sl], far[sl], w["step"], 0.0, -1, 0)
        sm = torch.empty(per, dtype=torch.int64, device=dev); a.sm_cnts = B.ptr(sm)
        rc = torch.empty(per, dtype=torch.int32, device=dev)
        runs = torch.empty((32, per), dtype=torch.int64, device=dev)
        meta = torch.zeros(4, dtype=torch.int64, device=dev)
        pk = torch.empty((per, 2), dtype=torch.int64, device=dev)
        scr = B.cumsum_scratch(per, dev)
        cap = 20_000_000 if chunks > 1 else 34_000_000
        ts = torch.empty(cap, dtype=torch.float32, device=dev); te = torch.empty_like(ts); ri = torch.empty(cap, dtype=torch.int64, device=dev)
        bufs.append((a, sm, rc, runs, meta, pk, scr, ts, te, ri, per, cap))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    evs = []
    for c, (a, sm, rc, runs, meta, pk, scr, ts, te, ri, per, cap) in enumerate(bufs):
        B.call("nfa_traverse_runs", C.byref(a), B.ptr(bits), B.ptr(rc), B.ptr(runs), 32, B.ptr(meta[3:4]), 0.0, None, 0, main.cuda_stream)
        st = side if overlap else main
        if overlap:
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        B.call("nfa_exclusive_cumsum_pairs_stats_i64", B.ptr(sm), per, B.ptr(pk), B.ptr(meta[0:3]), B.ptr(scr), st.cuda_stream)
        B.call("nfa_expand_runs", per, float(w["step"]), B.ptr(rc), B.ptr(runs), 32, B.ptr(pk), B.ptr(ts), B.ptr(te), None, B.ptr(ri), cap, st.cuda_stream)
    if overlap:
        ev = torch.cuda.Event(); ev.record(side); main.wait_event(ev)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for chunks, overlap in ((1, False), (2, False), (2, True), (4, True), (8, True)):
    ts = sorted(run(chunks, overlap) for _ in range(6))
    print(f"chunks {chunks} overlap {overlap!s:5s}: traversal wall us  min {ts[0]:7.1f}  median {ts[3]:7.1f}")
