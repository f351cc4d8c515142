#!/bin/bash
# usage: scripts/walk_sweep.sh "<flags1>" "<flags2>" ...   (each variant rebuilds the library with the flags)
out=gpurun_out/walk_sweep.jsonl
for f in "$@"; do
  export NERFACC_AMD_EXTRA_FLAGS="$f"
  python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  timeout -k 10 120 python scripts/walk_bench.py --tag="$f" 2>/dev/null | tee -a $out
  timeout -k 10 120 python scripts/walk_bench.py --tag="$f" --rays random 2>/dev/null | tee -a $out
  timeout -k 10 120 python scripts/walk_bench.py --tag="$f" --res 256 2>/dev/null | tee -a $out
done
