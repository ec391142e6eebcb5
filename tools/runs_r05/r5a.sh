#!/bin/bash
# round 5, first call: the sparse suffix table -- its own tests, then C4 read-derived 31-mers with the table off / on (counters, parity)
out=gpurun_out/r5a; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu > $out/sparse_tests.log 2>&1; rc=$?; echo "sparse tests rc=$rc"; tail -15 $out/sparse_tests.log
[ $rc -eq 0 ] || exit 1
for mode in 0 auto; do
  MSBWT_VERBOSE=1 MSBWT_SPARSE_TABLE=$mode timeout -k 10 500 python bench.py --workload c4 --query-kind reads --steps 10 --warmup 2 --counters --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/c4_sparse_$mode.json 2> $out/c4_sparse_$mode.err || { tail -5 $out/c4_sparse_$mode.err; exit 1; }
  grep -E "sparse table|load:" $out/c4_sparse_$mode.err | tail -12
  echo "c4 sparse=$mode $(python -c "import json;d=json.load(open('$out/c4_sparse_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity'], json.dumps(d['search_counters']), d['config'].get('index_bytes'))")"
done
