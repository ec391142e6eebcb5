#!/bin/bash
# round 5, last GPU minutes: the branch r6-deep-sparse-table (worktree .wt_deep: sparse table depths 25..28, NOT merged) -- its table tests against the
# oracle, then the metric's line with a depth-27 table
out=$GRAFT_REPO_ROOT/gpurun_out/r5deep; mkdir -p $out
cd .wt_deep || exit 1
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
timeout -k 10 240 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu -k "every_entry or counts_with" > $out/sparse_tests.log 2>&1; rc=$?
tail -4 $out/sparse_tests.log
if [ $rc -eq 0 ]; then
  timeout -k 10 300 python bench.py --sparse-depth 27 --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --counters --parity-sample 2000000 > $out/human_depth27.json 2> $out/human_depth27.log; rc=$?
  grep -v "bwt: group" $out/human_depth27.log | tail -5
  python - <<PY
import json
try:
    r = json.loads(open("$out/human_depth27.json").read().strip().splitlines()[-1]); sc = r["search_counters"]
    print("depth", r["config"]["sparse_table_depth"], "index GB %.1f" % (r["config"]["index_bytes"] / 1e9), "q/s %.4g" % (r["value"] or 0), "ms %.2f" % r["ms_per_step"], "lines/query %.3f" % sc["lines_per_query"], "parity", r["parity"])
    print(r["config"]["sparse_table"])
except Exception as e:
    print("no line", e)
PY
fi
kill $hb
exit $rc
