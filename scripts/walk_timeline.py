#!/usr/bin/env python3
"""Wave timeline of walk_kernel (build with NERFACC_AMD_EXTRA_FLAGS=-DNFA_WALK_STAMPS): how many wave slots are busy over the
launch, wave durations, what the last waves are.   python scripts/walk_timeline.py [--res 128]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--rays", default="image")
    ap.add_argument("--lpt", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    w = bench.make_workload(dev, 1024 * 1024, args.res, "shell10", args.rays, 0, "native")
    est = w["estimator"]
    n_waves = 1024 * 1024 // 64
    stamps = torch.zeros(n_waves * 4, dtype=torch.int64, device=dev)
    for _ in range(3):
        est._traverse(w["rays_o"], w["rays_d"], 0.0, 1e10, None, None, w["step"], False, 0.0)
    os.environ["NFA_WALK_STAMPS_PTR"] = str(stamps.data_ptr())
    est._traverse(w["rays_o"], w["rays_d"], 0.0, 1e10, None, None, w["step"], False, 0.0)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(n_waves, 4)
    if args.lpt:
        # the same launch with the tiles handed out longest first (the durations just measured: what an oracle order would give)
        d = (s[:, 1] - s[:, 0]).reshape(-1, 4).max(axis=1)
        if args.lpt.startswith("chord"):
            # what a pre-pass could know: the chord of one ray per tile through the box, in buckets, longest first, stable
            nb = int(args.lpt[5:] or 32)
            o = w["rays_o"][128::256].double(); dd = w["rays_d"][128::256].double()
            inv = 1.0 / dd
            t1 = (-1.0 - o) * inv; t2 = (1.0 - o) * inv
            tmin = torch.minimum(t1, t2).amax(-1).clamp_min(0.0); tmax = torch.maximum(t1, t2).amin(-1)
            chord = (tmax - tmin).clamp_min(0.0)
            key = (nb - 1 - (chord / chord.max() * (nb - 1e-6)).floor().clamp(0, nb - 1)).long()
            order = torch.sort(key, stable=True).indices.to(torch.int32).to(dev)
        elif args.lpt == "random":
            order = torch.randperm(len(d), dtype=torch.int32).to(dev)
        else:
            order = torch.from_numpy(np.argsort(-d, kind="stable").astype(np.int32)).to(dev)
        os.environ["NFA_WALK_ORDER_PTR"] = str(order.data_ptr())
        stamps.zero_()
        est._traverse(w["rays_o"], w["rays_d"], 0.0, 1e10, None, None, w["step"], False, 0.0)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(n_waves, 4)
    s = s[s[:, 0] > 0]
    tiles = (s[:, 3] >> 8)
    t0, t1 = s[:, 0].astype(np.float64), s[:, 1].astype(np.float64)
    base = t0.min()
    t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us
    dur = t1 - t0
    total = t1.max()
    # busy slots over time
    edges = np.linspace(0, total, 41)
    busy = [float(((t0 < b) & (t1 > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
    # per hardware wave slot: gaps between one wave's end and the next wave's start
    hw = s[:, 2].astype(np.int64); xcc = s[:, 3].astype(np.int64) & 15
    slot = (xcc << 20) | (hw & 0xFFFFF & ~(0xF << 16))   # wave, simd, pipe, cu, sh, se (tg_id masked)
    order = np.lexsort((t0, slot))
    ss, a0, a1 = slot[order], t0[order], t1[order]
    same = ss[1:] == ss[:-1]
    gaps = (a0[1:] - a1[:-1])[same]
    n_slots = len(np.unique(slot))
    simd = (xcc << 20) | (hw & 0xFFFF0 & ~(0xF << 16))
    n_simd = len(np.unique(simd))
    cu = (xcc << 20) | (hw & 0xFF00)
    per_xcc = [int((xcc == k).sum()) for k in range(8)]
    print(json.dumps({"slots_seen": n_slots, "simds_seen": n_simd, "cus_seen": len(np.unique(cu)), "waves_per_xcc": per_xcc,
                      "gap_us_mean": round(float(gaps.mean()), 2), "gap_us_p50": round(float(np.median(gaps)), 2),
                      "gap_us_p90": round(float(np.percentile(gaps, 90)), 2), "gap_us_max": round(float(gaps.max()), 2),
                      "gaps_sum_over_slots_us": round(float(gaps.sum() / n_slots), 1),
                      "first_start_spread_us": round(float(np.percentile(a0[np.r_[True, ~same]], 99)), 2)}))
    print(json.dumps({"res": args.res, "lpt": args.lpt, "kernel_us": round(total, 1), "wave_us_mean": round(dur.mean(), 1), "wave_us_p50": round(float(np.median(dur)), 1),
                      "wave_us_p95": round(float(np.percentile(dur, 95)), 1), "wave_us_max": round(dur.max(), 1),
                      "sum_wave_us_over_4096_slots": round(dur.sum() / 4096, 1), "waves": int(len(dur)), "tiles_per_wave_max": int(tiles.max()), "tiles_per_wave_min": int(tiles.min()),
                      "busy_waves_by_40ths": [int(b) for b in busy],
                      "start_us_p50": round(float(np.median(t0)), 1), "last_start_us": round(t0.max(), 1)}))


if __name__ == "__main__":
    main()
