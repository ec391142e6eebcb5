#!/bin/bash
# round 5, kernel revision 2 (three-instruction sparse scan; run blocks: the second bound continues the first bound's decode): parity, then every line again
out=gpurun_out/r5r; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python bench.py --blocks runs --queries 100000000 --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --no-cpu-baseline --counters --parity-sample 500000 > $out/human_runs.json 2> $out/human_runs.err || { tail -3 $out/human_runs.err; exit 1; }
echo "human run blocks k=31 $(python -c "import json;d=json.load(open('$out/human_runs.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
timeout -k 10 500 python bench.py --blocks runs --k 59 --queries 50000000 --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --no-cpu-baseline --parity-sample 200000 > $out/human_runs_k59.json 2> $out/human_runs_k59.err || { tail -3 $out/human_runs_k59.err; exit 1; }
echo "human run blocks k=59 $(python -c "import json;d=json.load(open('$out/human_runs_k59.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
timeout -k 10 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $out/c2.json 2> $out/c2.err || exit 1
echo "c2 $(python -c "import json;d=json.load(open('$out/c2.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
timeout -k 10 400 python bench.py --workload c3 --fused --steps 10 --warmup 2 --no-cpu-baseline --parity-sample 1000000 > $out/c3f.json 2> $out/c3f.err || exit 1
echo "c3 fused $(python -c "import json;d=json.load(open('$out/c3f.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
timeout -k 10 500 python bench.py --workload c4 --query-kind reads --steps 10 --warmup 2 --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/c4.json 2> $out/c4.err || exit 1
echo "c4 $(python -c "import json;d=json.load(open('$out/c4.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
timeout -k 10 500 python bench.py --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/human.json 2> $out/human.err || exit 1
echo "human $(python -c "import json;d=json.load(open('$out/human.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
