#!/bin/bash
# round 5, kernel revision 3 (index lines fetched non-temporally on indexes of 4 GiB and more): parity, then the lines
out=gpurun_out/r5y; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py tests/test_gpu_config_sizes.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for mode in auto 0; do
MSBWT_STREAM_LINES=$mode timeout -k 10 500 python bench.py --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/human_$mode.json 2> $out/human_$mode.err || exit 1
echo "human streaming=$mode rep$rep $(python -c "import json;d=json.load(open('$out/human_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
done; done
for mode in auto 0; do
MSBWT_STREAM_LINES=$mode timeout -k 10 500 python bench.py --blocks runs --queries 100000000 --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --no-cpu-baseline --parity-sample 500000 > $out/human_runs_$mode.json 2> $out/human_runs_$mode.err || exit 1
echo "human run blocks streaming=$mode $(python -c "import json;d=json.load(open('$out/human_runs_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
MSBWT_STREAM_LINES=$mode timeout -k 10 600 python bench.py --workload c4x3r --steps 5 --warmup 1 --no-cpu-baseline --stats-sample 200000 --parity-sample 500000 > $out/c4x3r_$mode.json 2> $out/c4x3r_$mode.err || exit 1
echo "c4x3r streaming=$mode $(python -c "import json;d=json.load(open('$out/c4x3r_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
done
