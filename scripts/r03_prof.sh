#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for rl in 6 4; do
export NERFACC_AMD_EXTRA_FLAGS="-DNFA_WALK_OP_RL=$rl"
python -c "from nerfacc_amd import _build; _build.build(force=True)" 2>/dev/null
for w in 256 512; do echo "== RL=$rl walk@96: concurrent $w"; NERFACC_AMD_EXPANDER_WGS=$w bash scripts/r03_trace.sh | tail -5 | grep "expand_units\|walk_publish"; done
done
