#!/bin/bash
# round 5, kernel revision 3: randomised parity soak with the non-temporal line loads forced on half of the configurations (new seeds)
out=gpurun_out/r5z; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for seed in 91 92 93; do
  STRESS_SEED=$seed timeout -k 10 330 python tools/stress_parity.py 200 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
