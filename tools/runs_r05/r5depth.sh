#!/bin/bash
# round 5, for the next round's plan: what a DEEPER sparse table would buy the metric.  Present k-mers of k = 27 / 25 / 23 on the depth-23 table need
# 1 bucket line + 2 / 1 / 0 pair steps -- the line counts k = 31 would have on a table of depth 27 / 29 / 31.  Same index, same kernel.
out=gpurun_out/r5depth; mkdir -p $out
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
rc=0
for k in 27 25 23; do
  timeout -k 10 400 python bench.py --k $k --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --counters --parity-sample 200000 > $out/k$k.json 2> $out/k$k.log || { rc=1; tail -5 $out/k$k.log; break; }
  python - <<PY
import json
r = json.loads(open("$out/k$k.json").read().strip().splitlines()[-1])
sc = r["search_counters"]
print("k", r["config"]["k"], "q/s %.4g" % r["value"], "ms %.2f" % r["ms_per_step"], "lines/query %.3f" % sc["lines_per_query"], "steps %.3f" % sc["steps_per_searched_query"], "rides", sc["raw"]["table_rides"], "parity", r["parity"]["mismatches"], "lines/s %.3g" % (r["value"] * sc["lines_per_query"]))
PY
done
kill $hb
exit $rc
