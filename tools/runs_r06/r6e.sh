#!/bin/bash
# round 6: the whole GPU suite on the tree with the two-tier table, the second sparse level and the sparse table behind run blocks; smoke(); one soak seed
out=gpurun_out/r6e; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 800 python -m pytest tests -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -4 $out/gputests.log
[ $rc -eq 0 ] || { grep -n "Error\|^E  \|FAILED" $out/gputests.log | head -40; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
for seed in ${SOAK_SEEDS:-901}; do
  STRESS_SEED=$seed timeout -k 10 260 python tools/stress_parity.py 150 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
