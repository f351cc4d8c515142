#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel of the library, from the compiler (-Rpass-analysis=kernel-resource-usage),
and for the kernels that spill: at which loop depth their scratch accesses sit (from the generated code).
    python scripts/resource_usage.py > profiles/r04_resource_usage.txt        (no GPU needed)"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfacc_amd import _build

def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))

print(f"# flags: {' '.join(_build.FLAGS)}")
print(f"# {'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>15s} {'waves/SIMD':>11s} {'LDS B/wg':>9s}")
for src in _build.SOURCES:
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([_build.hipcc(), *_build.FLAGS, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                            os.path.join(_build.CSRC, src), "-o", asm], capture_output=True, text=True)
        rows, cur = [], None
        for l in r.stderr.split("\n"):
            m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", l)
            if not m: continue
            if m.group(1) == "Function Name":
                cur = {"name": m.group(2)}; rows.append(cur)
            elif cur is not None:
                cur[m.group(1)] = m.group(2)
        seen, uniq = set(), []
        for row in rows:
            if row["name"] not in seen:
                seen.add(row["name"]); uniq.append(row)
        dm = demangle([row["name"] for row in uniq])
        print(f"## {src}")
        text = open(asm).read()
        for row in uniq:
            nm = re.sub(r"\(.*", "", dm.get(row["name"], row["name"])).replace("void ", "")[:100]
            print(f"  {nm:100s} {row.get('VGPRs','?'):>5s} {row.get('AGPRs','?'):>5s} {row.get('TotalSGPRs','?'):>5s} "
                  f"{row.get('ScratchSize [bytes/lane]','?'):>15s} {row.get('Occupancy [waves/SIMD]','?'):>11s} {row.get('LDS Size [bytes/block]','?'):>9s}")
            if row.get("ScratchSize [bytes/lane]", "0") != "0":
                m = re.search(r"^%s:.*?\.Lfunc_end" % re.escape(row["name"]), text, re.M | re.S)
                depth, cnt = 0, collections.Counter()
                for l in (m.group(0).split("\n") if m else []):
                    if l.startswith(".LBB"):
                        d = re.search(r"Depth=(\d+)", l); depth = int(d.group(1)) if d else 0
                    if "scratch_" in l: cnt[(depth, l.split()[0])] += 1
                    if "v_readlane" in l or "v_writelane" in l: cnt[(depth, "sgpr spill (" + l.split()[0] + ")")] += 1
                print("      scratch / scalar-spill instructions by loop depth (0 = outside every loop): " +
                      ", ".join(f"depth {d}: {n} x {op}" for (d, op), n in sorted(cnt.items())))
