#!/bin/bash
# round 5, for the next round's memory plan: the default human-scale line with pair blocks of stride 128 (1 B per symbol, 30 GB less than stride 96)
out=gpurun_out/r5stride; mkdir -p $out
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
MSBWT_PAIR_STRIDE=128 timeout -k 10 400 python bench.py --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --counters --parity-sample 200000 > $out/stride128.json 2> $out/stride128.log; rc=$?
kill $hb
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r5stride/stride128.json").read().strip().splitlines()[-1])
sc = r["search_counters"]
print("stride", r["config"]["pair_stride"], "index GB %.1f" % (r["config"]["index_bytes"] / 1e9), "q/s %.4g" % r["value"], "ms %.2f" % r["ms_per_step"], "lines/query %.3f" % sc["lines_per_query"], "second-line rate %.4f" % sc["second_line_rate"], "parity", r["parity"]["mismatches"])
PY
exit $rc
