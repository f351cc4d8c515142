#!/bin/bash
for cfg in "512,48" "1024,48" "1024,56" "2048,48" "768,48" "512,60"; do
  echo "NFA_REFILL=$cfg"; NFA_REFILL=$cfg timeout -k 10 300 python scripts/limit_sweep.py 4 2>/dev/null || exit 1
done
