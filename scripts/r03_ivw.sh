#!/bin/bash
for fl in "" "-DNFA_EXP_IV_WAVES=8" "-DNFA_EXP_IV_WAVES=7"; do
export NERFACC_AMD_EXTRA_FLAGS="$fl"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== '$fl'"; timeout -k 10 300 python scripts/api_traverse_probe.py 2>/dev/null | grep "expand" || exit 1
done
