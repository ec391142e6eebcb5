#!/bin/bash
# round 5, the two modes of the C4-sized lines (bounded: experiment 1 of 3): hipMalloc against physically contiguous memory for the pair blocks and
# the sparse table, instances side by side in one process -- plain, and under rocprofv3 (where round 4 saw the slow mode 29 times of 29);
# once on the round-4 index (no sparse table) and once on today's default
out=$PWD/gpurun_out/r5g; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for sparse in 0 auto; do
  MSBWT_VERBOSE=1 timeout -k 10 400 python tools/alloc_probe.py c4r 3 2 $sparse > $out/plain_sparse_$sparse.log 2> $out/plain_sparse_$sparse.err || { tail -5 $out/plain_sparse_$sparse.err; exit 1; }
  echo "== plain, sparse table $sparse"; cat $out/plain_sparse_$sparse.log; grep -c "physically contiguous" $out/plain_sparse_$sparse.err
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/alloc_probe.py c4r 3 1 auto > $out/rocprof_sparse_auto.log 2> $out/rocprof_sparse_auto.err || { tail -5 $out/rocprof_sparse_auto.err; exit 1; }
echo "== under rocprofv3 --kernel-trace, sparse table auto"; cat $out/rocprof_sparse_auto.log
rm -rf $out/prof
