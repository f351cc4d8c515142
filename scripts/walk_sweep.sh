#!/bin/bash
# usage: scripts/walk_sweep.sh [-q] "<flags1>" "<flags2>" ...   (each variant rebuilds the library with the flags; -q: cfg 2 image rays only)
out=gpurun_out/walk_sweep.jsonl
quick=0
if [ "$1" = "-q" ]; then quick=1; shift; fi
for f in "$@"; do
  export NERFACC_AMD_EXTRA_FLAGS="$f"
  python -c "from nerfacc_amd import _build; _build.build()" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  timeout -k 10 120 python scripts/walk_bench.py --tag="$f" 2>/dev/null | tee -a $out
  if [ $quick = 0 ]; then
    timeout -k 10 120 python scripts/walk_bench.py --tag="$f" --rays random 2>/dev/null | tee -a $out
    timeout -k 10 120 python scripts/walk_bench.py --tag="$f" --res 256 2>/dev/null | tee -a $out
  fi
done
