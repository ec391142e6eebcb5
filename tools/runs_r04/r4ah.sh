#!/bin/bash
# the reworked lanes kernel (packed queries as instantiations of their own; side-array entries fetched in step A): whole GPU suite + a parity soak
out=gpurun_out/r4ah; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -4 $out/gputests.log
[ $rc = 0 ] || exit 1
STRESS_SEED=41 timeout -k 10 400 python tools/stress_parity.py 240 > $out/stress.log 2>&1; rc=$?; echo "stress rc=$rc"; tail -2 $out/stress.log
