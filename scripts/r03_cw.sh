#!/bin/bash
for fl in "" "-DNFA_CONE_WALK_WAVES=6 -DNFA_CONE_REFILL_WAVES=6" "-DNFA_CONE_WALK_WAVES=4 -DNFA_CONE_REFILL_WAVES=4" "-DNFA_CONE_WALK_WAVES=8 -DNFA_CONE_REFILL_WAVES=8"; do
export NERFACC_AMD_EXTRA_FLAGS="$fl"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== '$fl'"; timeout -k 10 300 python scripts/limit_sweep.py 4 0 2>/dev/null || exit 1
done
