#!/bin/bash
# round 4: soak over the FINAL sources (the ordering passes now stage their tiles in LDS; run blocks built on the device)
out=gpurun_out/r4x; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for seed in 61 62 63; do
  STRESS_SEED=$seed timeout -k 10 330 python tools/stress_parity.py 240 > $out/soak_seed$seed.log 2>&1; echo "soak seed $seed rc=$?"; tail -1 $out/soak_seed$seed.log
done
