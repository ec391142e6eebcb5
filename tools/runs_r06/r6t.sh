#!/bin/bash
# round 6: a longer soak over the final sources (four more seeds)
out=gpurun_out/r6t; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
for seed in 501 502 503 504; do
  STRESS_SEED=$seed timeout -k 10 260 python tools/stress_parity.py 180 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
