#!/bin/bash
# round 5: rocprofv3 evidence for the repeat-genome line (profiles/r05_v3: kernel stats, FETCH_SIZE, WRITE_SIZE, L2 hits), then the same index
# with the sparse table switched off (the round-4 index: packed direct table, depth 17) for the comparison
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
PROF_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum" bash tools/profile_bench.sh r05_v3 human_repeats --genome repeats 2>&1 | tail -3; rc=$?
if [ $rc -eq 0 ]; then
  mkdir -p gpurun_out/r5rep2
  MSBWT_SPARSE_TABLE=0 timeout -k 10 600 python bench.py --genome repeats --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --counters --parity-sample 200000 \
      > gpurun_out/r5rep2/bench_direct_table.json 2> gpurun_out/r5rep2/bench_direct_table.log; rc=$?
  grep -v "bwt: group" gpurun_out/r5rep2/bench_direct_table.log | tail -8
fi
kill $hb
exit $rc
