#!/bin/bash
# round 5, the two modes (experiment 2 of 3): is it the small, hot superblock table (155 lines in 5 pages on a C4-sized index: few L2 channels
# serve all its loads, and which ones depends on the pages' physical place)?  1 copy against 32 copies on pages of their own, instances side by side.
out=$PWD/gpurun_out/r5h; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py -x -q -m gpu -k "sparse or lanes" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
for sparse in auto 0; do
  timeout -k 10 400 python tools/alloc_probe.py c4r 4 2 $sparse malloc:1,malloc:32 > $out/copies_sparse_$sparse.log 2> $out/copies_sparse_$sparse.err || { tail -5 $out/copies_sparse_$sparse.err; exit 1; }
  echo "== sparse table $sparse"; cat $out/copies_sparse_$sparse.log
done
