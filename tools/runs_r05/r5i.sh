#!/bin/bash
# round 5, the two modes (experiment 3 of 3): does a one-millisecond gather of random lines over an array predict the speed of the launches that
# use it, and does keeping the best-served of 6 candidate allocations (pair blocks, sparse table) make every instance a fast one?
out=$PWD/gpurun_out/r5i; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
MSBWT_VERBOSE=1 timeout -k 10 500 python tools/alloc_probe.py c4r 5 2 auto malloc:0:1,malloc:0:6 > $out/candidates.log 2> $out/candidates.err || { tail -5 $out/candidates.err; exit 1; }
cat $out/candidates.log; grep "placement of" $out/candidates.err
