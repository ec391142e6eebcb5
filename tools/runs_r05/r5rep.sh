#!/bin/bash
# round 5: the headline's index over a genome WITH repeats (bench.py --genome repeats: synth.repeat_genome at human scale, exact MSBWT by
# synth/bwt_reads.py msbwt_rle_repeats) -- present 31-mers drawn like read windows, parity against the oracle, search counters, one live
# FETCH_SIZE pass.  A lab line beside the metric's repeat-free one.
out=gpurun_out/r5rep; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
timeout -k 10 1080 python bench.py --genome repeats --no-c4 --no-c5 --no-sorted --counters $R5REP_EXTRA > $out/bench.json 2> $out/bench.log; rc=$?
kill $hb
grep -v "^\[bench\] rank 0: bwt: group" $out/bench.log | tail -45
echo "rc=$rc"
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r5rep/bench.json").read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ("value", "ms_per_step")}, r.get("roofline", {}).get("frac"), r.get("roofline", {}).get("traffic_source"))
    print(r.get("search_counters"))
    print(r["config"].get("sparse_table"), r["config"].get("index_bytes"))
except Exception as e:
    print("no line:", e)
PY
exit $rc
