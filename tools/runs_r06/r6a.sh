#!/bin/bash
# round 6, first call on the merged tree: the whole GPU suite, smoke(), then the default bench run -- its stdout line must be compact and parse
out=gpurun_out/r6a; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -4 $out/gputests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $out/gputests.log | tail -20; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
t0=$(date +%s)
MSBWT_VERBOSE=1 timeout -k 10 420 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; rc=$?
echo "bench rc=$rc in $(( $(date +%s) - t0 )) s, stdout bytes: $(wc -c < $out/bench_default.json)"
cp bench_extras.json $out/bench_extras.json 2>/dev/null
cat $out/bench_default.json
grep -n "variant lines\|PARITY\|sparse table: depth\|Traceback" $out/bench_default.err | tail -20
exit $rc
