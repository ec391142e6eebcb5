#!/bin/bash
# round 6: the driver's own bench command once more (the variant lines rebuild their tables before the batch returns to HBM)
out=gpurun_out/r6y; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_bench_smoke.py -x -q -m gpu -k "default_run or contract" > $out/smoke_tests.log 2>&1; rc=$?; echo "bench smoke rc=$rc"; tail -2 $out/smoke_tests.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/smoke_tests.log | head -20; exit $rc; }
t0=$(date +%s)
MSBWT_VERBOSE=1 timeout -k 10 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; rc=$?
echo "bench rc=$rc in $(( $(date +%s) - t0 )) s, stdout bytes: $(wc -c < $out/bench_default.json)"
cp bench_extras.json $out/bench_extras.json 2>/dev/null
python -c "import json;c=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]);print(c['value'], c['roofline']['frac'], c['parity'], {k:v for k,v in c['extras'].items() if 'short_k' in k or 'undeclared' in k or 'sparse_off' in k})"
grep -n "variant lines\|PARITY\|Traceback\|second level" $out/bench_default.err | tail -12
exit $rc
