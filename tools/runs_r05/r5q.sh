#!/bin/bash
# round 5: randomised parity soak over the final sources (sparse table depths, run blocks with k > 32, budgets, ordering, packed queries ...)
out=gpurun_out/r5q; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for seed in 81 82 83; do
  STRESS_SEED=$seed timeout -k 10 400 python tools/stress_parity.py 240 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
