#!/bin/bash
# round 5: the repeat-genome line's cost by copy number at human scale (copy_number_bins of the main line: 1-99 ... >= 10^6 occurrences)
out=gpurun_out/r5rep3; mkdir -p $out
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
timeout -k 10 900 python bench.py --genome repeats --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --counters --parity-sample 200000 > $out/bench.json 2> $out/bench.log; rc=$?
kill $hb
grep -v "bwt: group" $out/bench.log | tail -6
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r5rep3/bench.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["parity"])
for b in r["copy_number_bins"]["bins"]:
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in b.items()})
PY
exit $rc
