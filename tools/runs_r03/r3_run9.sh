#!/bin/bash
# round-3 GPU call 9: randomised parity soak on the final sources (two seeds, 5 minutes each)
set -o pipefail
O=gpurun_out/r3n; mkdir -p $O
STRESS_SEED=31 python tools/stress_parity.py 300 2>&1 | tee $O/soak_seed31.log | tail -3 &&
STRESS_SEED=32 python tools/stress_parity.py 300 2>&1 | tee $O/soak_seed32.log | tail -3
