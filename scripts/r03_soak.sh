#!/bin/bash
for seed in 1 2 3 4 5 6; do
  NFA_CONE_SEED=$seed NFA_FUZZ_SEED=$((seed + 100)) timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cone_walk_forms or traversal_fuzz" 2>&1 | tail -1 || exit 1
done
