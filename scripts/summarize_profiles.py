#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>_{trace,fetch,write} (rocprofv3 CSVs) into profiles/<tag>_*.{csv,json}.

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE reports half
the bytes of wide coalesced loads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte
stores.  The figures are per launch, averaged over the launches of the run.
"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    assert f, pattern
    return max(f, key=os.path.getmtime)   # gpurun merges a call's files into the directory: earlier collections' files stay


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "bench_field::").split("(")[0][:110]


stats = list(csv.DictReader(open(one(f"{tag}_trace/*/*_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "percent"])
    for r in stats:
        w.writerow([short(r["Name"]), r["Calls"], f"{float(r['AverageNs']) / 1e3:.2f}", f"{float(r['MinNs']) / 1e3:.2f}",
                    f"{float(r['MaxNs']) / 1e3:.2f}", f"{float(r['TotalDurationNs']) / 1e6:.3f}", r["Percentage"]])


def counter(kind, name):
    rows = csv.DictReader(open(one(f"{tag}_{kind}/*/*_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == name:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in stats}
out = {}
for k in sorted(set(fetch) | set(write)):
    if "nfa::" not in k:
        continue
    fb, wb = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    out[k] = {"avg_us": avg_us.get(k), "FETCH_SIZE_bytes_raw": fb, "WRITE_SIZE_bytes": wb,
              "hbm_bytes_per_launch": 2 * fb + wb,
              "hbm_GBps": (2 * fb + wb) / (avg_us[k] * 1e-6) / 1e9 if avg_us.get(k) else None}
# the content hash of the library sources the counters were taken with (collect_profiles.sh); bench.py quotes the traffic
# only while the library it runs still has this hash
hash_file = os.path.join(src, f"{tag}_source_hash.txt")
out["_library_source_hash"] = open(hash_file).read().strip().splitlines()[-1] if os.path.exists(hash_file) else None
json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
out.pop("_library_source_hash")
print(f"wrote profiles/{tag}_kernel_stats.csv and profiles/{tag}_hbm_traffic.json")
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["avg_us"] or 0))[:12]:
    print(f"  {v['avg_us']:8.1f} us  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  {v['hbm_GBps']:8.1f} GB/s  {k[:70]}")

# secondary configurations: kernel-trace stats only
for cfg in ("cfg2_compacting", "cfg2_random", "cfg3", "cfg5", "cfg4_256"):
    f = glob.glob(os.path.join(src, f"{tag}_{cfg}_trace/*/*_kernel_stats.csv"))
    if not f:
        continue
    rows = list(csv.DictReader(open(max(f, key=os.path.getmtime))))
    with open(os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['AverageNs']) / 1e3:.2f}", f"{float(r['MinNs']) / 1e3:.2f}",
                        f"{float(r['MaxNs']) / 1e3:.2f}", f"{float(r['TotalDurationNs']) / 1e6:.3f}", r["Percentage"]])
    print(f"wrote profiles/{tag}_{cfg}_kernel_stats.csv")
