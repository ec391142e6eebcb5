#!/bin/bash
# round 6: soak over the FINAL kernel sources (one kernel per table layout, the pinned wait), three seeds; then the bench smoke test that reads the new extras
out=gpurun_out/r6s; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
for seed in 301 302 303; do
  STRESS_SEED=$seed timeout -k 10 200 python tools/stress_parity.py 120 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
timeout -k 10 300 python -m pytest tests/test_bench_smoke.py -x -q -m gpu -k "default_run" > $out/smoke_tests.log 2>&1; rc=$?; echo "bench smoke rc=$rc"; tail -2 $out/smoke_tests.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/smoke_tests.log | head -20; exit $rc; }
