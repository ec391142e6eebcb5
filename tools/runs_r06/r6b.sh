#!/bin/bash
# round 6: rocprofv3 evidence for the kernels of the merged tree (profiles/r06_v1; afterwards tools/profile_collect.sh r06_v1 <name> for each).
# Two calls: "r6b.sh big" = human, human_runs, human_repeats;  "r6b.sh small" = c4, c4r, c2, c3_fused
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
if [ "$1" = big ]; then
  bash tools/profile_bench.sh r06_v1 human 2>&1 | tail -2 &&
  PROF_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum" bash tools/profile_bench.sh r06_v1 human_runs --blocks runs --queries 100000000 2>&1 | tail -2 &&
  PROF_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum" bash tools/profile_bench.sh r06_v1 human_repeats --genome repeats 2>&1 | tail -2
else
  bash tools/profile_bench.sh r06_v1 c4_reads --workload c4 2>&1 | tail -2 &&
  bash tools/profile_bench.sh r06_v1 c4r_reads --workload c4r 2>&1 | tail -2 &&
  bash tools/profile_bench.sh r06_v1 c2 --workload c2 2>&1 | tail -2 &&
  bash tools/profile_bench.sh r06_v1 c3_fused --workload c3 --fused 2>&1 | tail -2
fi
rc=$?
kill $hb
exit $rc
