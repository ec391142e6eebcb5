#!/bin/bash
# round 6: the N > 1 step's exchange as a gather to rank 0 (default) or an all_gather -- the bench smoke tests (gloo world 2 on the shared GPU, real RCCL on one rank)
out=gpurun_out/r6j; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 800 python -m pytest tests/test_bench_smoke.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?; echo "bench smoke tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED\|Error" $out/tests.log | head -40; exit $rc; }
