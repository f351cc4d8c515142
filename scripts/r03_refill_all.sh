#!/bin/bash
echo "unlimited, one ray per lane (binned when it pays)"; timeout -k 10 300 python scripts/limit_sweep.py 0 2>/dev/null || exit 1
for cfg in "512,48" "256,48" "128,48" "512,32"; do
for b in "" 0; do
echo "unlimited, refill $cfg bin='$b'"; NFA_LS_BIN=$b NFA_REFILL=$cfg NFA_REFILL_ALL=1 timeout -k 10 300 python scripts/limit_sweep.py 0 2>/dev/null || exit 1
done; done
