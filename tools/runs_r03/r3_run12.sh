#!/bin/bash
# round-3 GPU call 12: longer randomised parity soak on the final sources (three more seeds, 5 minutes each)
set -o pipefail
O=gpurun_out/r3q; mkdir -p $O
for S in 41 42 43; do STRESS_SEED=$S python tools/stress_parity.py 300 2>&1 | tee $O/soak_seed$S.log | tail -1 || exit 1; done
