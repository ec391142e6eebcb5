#!/bin/bash
# round 4: run blocks built on the device -- parity (runs mode + builder test), then the load time and rate at human scale
out=gpurun_out/r4u; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "runs or run_block" > $out/parity.log 2>&1; rc=$?; echo "parity(runs) rc=$rc"; tail -4 $out/parity.log
[ $rc -eq 0 ] || exit 1
for build in device host; do
MSBWT_BUILD=$build MSBWT_VERBOSE=1 timeout -k 10 600 python bench.py --blocks runs --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --queries 100000000 --no-cpu-baseline --stats-sample 100000 --parity-sample 500000 > $out/human_runs_$build.json 2> $out/human_runs_$build.err || exit 1
echo "human runs build=$build $(python -c "import json;d=json.load(open('$out/human_runs_$build.json'));print('%.4g q/s %.2f ms' % (d['value'], d['ms_per_step']), d['parity'], d['config']['index_bytes'])")"
grep -E "symbols on the GPU|load:" $out/human_runs_$build.err | cut -c1-160
done
