#!/bin/bash
# round 6, final validation of the committed tree: the whole GPU suite, smoke(), the driver's own bench command; everything kept under gpurun_out/r6z
out=gpurun_out/r6z; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 800 python -m pytest tests -q -m gpu > $out/gpu_suite.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 $out/gpu_suite.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/gpu_suite.log | head -40; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
t0=$(date +%s)
MSBWT_VERBOSE=1 timeout -k 10 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; rc=$?
echo "bench rc=$rc in $(( $(date +%s) - t0 )) s, stdout bytes: $(wc -c < $out/bench_default.json)"
cp bench_extras.json $out/bench_extras.json 2>/dev/null
cat $out/bench_default.json
grep -n "variant lines\|PARITY\|Traceback\|second level" $out/bench_default.err | tail -12
exit $rc
