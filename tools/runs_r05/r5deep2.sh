#!/bin/bash
# round 5, last GPU minutes: more of the branch r6-deep-sparse-table (worktree .wt_deep) -- all of its sparse-table tests, a short randomised soak with
# depths 25 / 27 / 28 among the settings, and the repeat-genome line with a depth-27 table
out=$GRAFT_REPO_ROOT/gpurun_out/r5deep; mkdir -p $out
cd .wt_deep || exit 1
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
timeout -k 10 200 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu > $out/sparse_tests_all.log 2>&1; rc=$?
tail -2 $out/sparse_tests_all.log
if [ $rc -eq 0 ]; then STRESS_SEED=101 timeout -k 10 150 python tools/stress_parity.py 75 > $out/soak_seed101.log 2>&1; rc=$?; tail -1 $out/soak_seed101.log; fi
if [ $rc -eq 0 ]; then
  timeout -k 10 200 python bench.py --genome repeats --sparse-depth 27 --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --no-oracle --counters > $out/repeats_depth27.json 2> $out/repeats_depth27.log; rc=$?
  python - <<PY
import json
try:
    r = json.loads(open("$out/repeats_depth27.json").read().strip().splitlines()[-1]); sc = r["search_counters"]
    print("repeats depth", r["config"]["sparse_table_depth"], "index GB %.1f" % (r["config"]["index_bytes"] / 1e9), "q/s %.4g" % (r["value"] or 0), "ms %.2f" % r["ms_per_step"], "lines/query %.3f" % sc["lines_per_query"])
    for b in r["copy_number_bins"]["bins"]: print(b.get("count_from"), b.get("value"), b.get("lines_per_query"), b.get("counts_equal_main_run"))
except Exception as e:
    print("no line", e)
PY
fi
kill $hb
exit $rc
