#!/bin/bash
# round 6: the two-tier table's own bucket bound (probe limit 3: half the complete table's least size) -- its tests, the C4-sized test, one soak seed, the default bench
out=gpurun_out/r6u; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 500 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu > $out/sparse.log 2>&1; rc=$?; echo "sparse tests rc=$rc"; tail -2 $out/sparse.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/sparse.log | head -30; exit $rc; }
timeout -k 10 400 python -m pytest tests/test_gpu_config_sizes.py -x -q -m gpu -k "c4_real" > $out/c4.log 2>&1; rc=$?; echo "c4 test rc=$rc"; tail -2 $out/c4.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/c4.log | head -30; exit $rc; }
STRESS_SEED=601 timeout -k 10 200 python tools/stress_parity.py 120 > $out/soak_seed601.log 2>&1; rc=$?; echo "seed 601 rc=$rc: $(tail -1 $out/soak_seed601.log)"
[ $rc -eq 0 ] || { tail -20 $out/soak_seed601.log; exit 1; }
t0=$(date +%s)
MSBWT_VERBOSE=1 timeout -k 10 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; rc=$?
echo "bench rc=$rc in $(( $(date +%s) - t0 )) s, stdout bytes: $(wc -c < $out/bench_default.json)"
cp bench_extras.json $out/bench_extras.json 2>/dev/null
python -c "import json;c=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]);print(c['value'], c['roofline']['frac'], c['parity'], {k:v for k,v in c['extras'].items() if 'budgeted' in k or 'short_k' in k})"
grep -n "two-tier\|PARITY\|Traceback" $out/bench_default.err | tail -8
exit $rc
