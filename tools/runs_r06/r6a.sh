#!/bin/bash
# round 6, first call after merging r6-deep-sparse-table: the whole GPU suite, smoke(), a soak over three seeds (sparse depths 25 / 27 / 28 among the settings)
out=gpurun_out/r6a; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -4 $out/gputests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
for seed in 111 112 113; do
  STRESS_SEED=$seed timeout -k 10 260 python tools/stress_parity.py 150 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
