#!/bin/bash
mkdir -p gpurun_out
run() { timeout -k 10 500 python bench.py --only cfg5_testmode --steps 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['cfg5_testmode']['test_mode_loop']; print(round(d['ms_per_image'],2), d['total_samples'])"; }
echo "== default"; run || exit 1
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_SEG_TILE_ROWS=1024"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== rows 1024, tile 1024"; run || exit 1
echo "== rows 1024, tile 4096"; NFA_SEG_TILE=4096 run || exit 1
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_SEG_TILE_ROWS=64"
python -c "from nerfacc_amd import _build; _build.build(force=True)" > /dev/null 2>&1
echo "== rows 64, tile 1024"; run || exit 1
