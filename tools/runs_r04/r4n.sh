#!/bin/bash
# round 4: run blocks on the lanes kernel -- parity in every mode, then what it brings
out=gpurun_out/r4n; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -6 $out/parity.log
[ $rc -eq 0 ] || exit 1
for search in lanes groups; do
  MSBWT_SEARCH=$search timeout -k 10 400 python bench.py --workload c4 --query-kind reads --blocks runs --steps 5 --warmup 1 --no-cpu-baseline --stats-sample 100000 --parity-sample 200000 > $out/c4_runs_$search.json 2> $out/c4_runs_$search.err || exit 1
  echo "c4 runs search=$search $(python -c "import json;d=json.load(open('$out/c4_runs_$search.json'));print('%.4g q/s %.2f ms' % (d['value'], d['ms_per_step']), d['parity'], d['config']['index_bytes'], d['config']['table_depth'])")"
done
for search in lanes groups; do
  MSBWT_SEARCH=$search timeout -k 10 600 python bench.py --blocks runs --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --queries 100000000 --no-cpu-baseline --stats-sample 100000 --parity-sample 200000 > $out/human_runs_$search.json 2> $out/human_runs_$search.err || exit 1
  echo "human runs search=$search $(python -c "import json;d=json.load(open('$out/human_runs_$search.json'));print('%.4g q/s %.2f ms' % (d['value'], d['ms_per_step']), d['parity'], d['config']['index_bytes'], d['config']['table_depth'])")"
done
