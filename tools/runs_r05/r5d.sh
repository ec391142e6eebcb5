#!/bin/bash
# round 5: sparse-table lookups riding along in free second-line slots -- parity, then C4 and the human-scale line
out=gpurun_out/r5d; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py -x -q -m gpu -k "sparse or auto" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log
[ $rc -eq 0 ] || exit 1
MSBWT_VERBOSE=1 timeout -k 10 500 python bench.py --workload c4 --query-kind reads --steps 10 --warmup 2 --counters --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/c4.json 2> $out/c4.err || { tail -5 $out/c4.err; exit 1; }
echo "c4 $(python -c "import json;d=json.load(open('$out/c4.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity'], json.dumps(d['search_counters']['raw']))")"
MSBWT_VERBOSE=1 timeout -k 10 500 python bench.py --workload c4r --steps 10 --warmup 2 --counters --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/c4r.json 2> $out/c4r.err || { tail -5 $out/c4r.err; exit 1; }
grep -E "sparse table" $out/c4r.err | tail -3
echo "c4r $(python -c "import json;d=json.load(open('$out/c4r.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity'], json.dumps(d['search_counters']['raw']))")"
timeout -k 10 500 python bench.py --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 --counters --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/human.json 2> $out/human.err || { tail -5 $out/human.err; exit 1; }
echo "human $(python -c "import json;d=json.load(open('$out/human.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity'], json.dumps(d['search_counters']['raw']), d['search_counters']['lines_per_query'])")"
