#!/bin/bash
# round 6: the two run-block tests whose expectations changed, a soak over three seeds, then run blocks behind a sparse table at human scale (r6d)
out=gpurun_out/r6f; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "run_block" > $out/runblock_tests.log 2>&1; rc=$?; echo "run-block tests rc=$rc"; tail -2 $out/runblock_tests.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/runblock_tests.log | head -20; exit $rc; }
for seed in 201 202 203; do
  STRESS_SEED=$seed timeout -k 10 200 python tools/stress_parity.py 110 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
bash tools/runs_r06/r6d.sh
