#!/bin/bash
echo "adaptive chunk"; timeout -k 10 300 python scripts/testmode_iters.py | tail -14
for cfg in "512,48" "256,48"; do echo "NFA_REFILL=$cfg"; NFA_REFILL=$cfg timeout -k 10 300 python scripts/testmode_iters.py | tail -14; done
