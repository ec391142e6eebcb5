#!/bin/bash
out=gpurun_out/r4o; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "runs" > $out/parity.log 2>&1; rc=$?; echo "parity(runs) rc=$rc"; tail -4 $out/parity.log
[ $rc -eq 0 ] || exit 1
MSBWT_SEARCH=lanes timeout -k 10 400 python bench.py --workload c4 --query-kind reads --blocks runs --steps 5 --warmup 1 --no-cpu-baseline --stats-sample 100000 --parity-sample 200000 --counters > $out/c4_runs_lanes.json 2> $out/c4_runs_lanes.err || exit 1
echo "c4 runs lanes $(python -c "import json;d=json.load(open('$out/c4_runs_lanes.json'));print('%.4g q/s %.2f ms' % (d['value'], d['ms_per_step']), d['parity']['mismatches'], json.dumps(d['search_counters']['raw']))")"
MSBWT_SEARCH=lanes timeout -k 10 600 python bench.py --blocks runs --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --queries 100000000 --no-cpu-baseline --stats-sample 100000 --parity-sample 200000 --counters > $out/human_runs_lanes.json 2> $out/human_runs_lanes.err || exit 1
echo "human runs lanes $(python -c "import json;d=json.load(open('$out/human_runs_lanes.json'));print('%.4g q/s %.2f ms' % (d['value'], d['ms_per_step']), d['parity']['mismatches'], json.dumps(d['search_counters']['raw']))")"
