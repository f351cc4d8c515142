#!/usr/bin/env python3
"""Times the kernels of the sampler's traversal alone (walk + cumsum + expansion) with HIP events.
    python scripts/walk_bench.py [--res 128] [--rays image|random] [--reps 20] [--bin-rays]
Build-time knobs for experiments go through NERFACC_AMD_EXTRA_FLAGS (e.g. -DNFA_WALK_WAVES=8)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--rays", default="image")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bin-rays", action="store_true")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    w = bench.make_workload(dev, 1024 * 1024, args.res, "shell10", args.rays, 0, "native")
    est = w["estimator"]
    est.bin_rays = bool(args.bin_rays)
    for _ in range(3):
        out = est._traverse(w["rays_o"], w["rays_d"], 0.0, 1e10, None, None, w["step"], False, 0.0)
    timer = bench.KernelTimer(); timer.install()
    for _ in range(args.reps):
        out = est._traverse(w["rays_o"], w["rays_d"], 0.0, 1e10, None, None, w["step"], False, 0.0)
    ks = timer.summary(args.reps); timer.uninstall()
    print(json.dumps({"tag": args.tag, "flags": os.environ.get("NERFACC_AMD_EXTRA_FLAGS", ""), "res": args.res, "rays": args.rays,
                      "samples": int(out[0].numel()),
                      "us": {k: round(v["ms_per_launch"] * 1e3, 1) for k, v in ks.items()}}))


if __name__ == "__main__":
    main()
