#!/bin/bash
# round 6: rocprofv3 evidence for the final kernel sources (profiles/r06_v1; afterwards tools/profile_collect.sh r06_v1 <name> for each).
#   r6p.sh big1  = human (k = 31 declared: depth 31), human_k_unknown (depth 23 + the second level)
#   r6p.sh big2  = human_runs (run blocks behind the depth-23 table, 1e8 queries), human_repeats (the repeat-bearing genome)
#   r6p.sh small = c4_reads (k declared), c4_k_unknown, c4_two_tier (depth 23, two-tier forced), c4_random, c4r_reads, c2, c3_fused
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
trap "kill $hb" EXIT
LIGHT="stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum"
case "$1" in
big1)
  bash tools/profile_bench.sh r06_v1 human 2>&1 | tail -2 &&
  bash tools/profile_bench.sh r06_v1 human_k_unknown --query-length-hint 0 2>&1 | tail -2 ;;
big2)
  PROF_PASSES="$LIGHT SQ_WAVE_CYCLES" bash tools/profile_bench.sh r06_v1 human_runs --blocks runs --queries 100000000 --query-length-hint 0 2>&1 | tail -2 &&
  PROF_PASSES="$LIGHT" bash tools/profile_bench.sh r06_v1 human_repeats --genome repeats 2>&1 | tail -2 ;;
small)   # the C4-sized lines on READ-DERIVED 31-mers (what the two-tier form is about) and on BASELINE configs[3]'s random ones, then C2 and C3 fused
  bash tools/profile_bench.sh r06_v1 c4_reads --workload c4 --query-kind reads 2>&1 | tail -2 &&
  PROF_PASSES="$LIGHT SQ_WAVE_CYCLES" bash tools/profile_bench.sh r06_v1 c4_two_tier --workload c4 --query-kind reads --query-length-hint 0 --sparse-tiers 1 2>&1 | tail -2 &&
  PROF_PASSES="$LIGHT SQ_WAVE_CYCLES" bash tools/profile_bench.sh r06_v1 c4_k_unknown --workload c4 --query-kind reads --query-length-hint 0 2>&1 | tail -2 &&
  PROF_PASSES="$LIGHT" bash tools/profile_bench.sh r06_v1 c4_random --workload c4 2>&1 | tail -2 &&
  PROF_PASSES="$LIGHT" bash tools/profile_bench.sh r06_v1 c4r_reads --workload c4r --query-kind reads 2>&1 | tail -2 &&
  bash tools/profile_bench.sh r06_v1 c2 --workload c2 2>&1 | tail -2 &&
  bash tools/profile_bench.sh r06_v1 c3_fused --workload c3 --fused 2>&1 | tail -2 ;;
*) echo "usage: r6p.sh big1|big2|small"; exit 2 ;;
esac
