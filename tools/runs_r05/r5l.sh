#!/bin/bash
# round 5: the other BASELINE configs with the new defaults (C2 random 21-mers use the direct table, now depth 15; C3 fused and the run-block
# format with k > 32 on the lanes kernel), against the same with the sparse table off
out=gpurun_out/r5l; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "runs" > $out/tests_runs.log 2>&1; rc=$?; echo "runs-mode tests rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" $out/tests_runs.log | tail -5
for mode in auto 0; do
  MSBWT_SPARSE_TABLE=$mode timeout -k 10 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $out/c2_$mode.json 2> $out/c2_$mode.err || { tail -3 $out/c2_$mode.err; exit 1; }
  echo "c2 sparse=$mode $(python -c "import json;d=json.load(open('$out/c2_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'], d['config']['index_bytes'], d['config']['table_depth'])")"
  MSBWT_SPARSE_TABLE=$mode timeout -k 10 400 python bench.py --workload c3 --fused --steps 10 --warmup 2 --no-cpu-baseline --parity-sample 1000000 > $out/c3f_$mode.json 2> $out/c3f_$mode.err || { tail -3 $out/c3f_$mode.err; exit 1; }
  echo "c3 fused sparse=$mode $(python -c "import json;d=json.load(open('$out/c3f_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'], d['config']['index_bytes'], d['config']['table_depth'])")"
  MSBWT_SPARSE_TABLE=$mode timeout -k 10 400 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --parity-sample 1000000 > $out/c3_$mode.json 2> $out/c3_$mode.err || { tail -3 $out/c3_$mode.err; exit 1; }
  echo "c3 matrix sparse=$mode $(python -c "import json;d=json.load(open('$out/c3_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'])")"
done
timeout -k 10 500 python bench.py --blocks runs --queries 100000000 --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --no-cpu-baseline --counters --parity-sample 500000 > $out/human_runs.json 2> $out/human_runs.err || { tail -3 $out/human_runs.err; exit 1; }
echo "human run blocks k=31 $(python -c "import json;d=json.load(open('$out/human_runs.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'], d['config']['index_bytes'], d['search_counters']['lines_per_query'])")"
timeout -k 10 500 python bench.py --blocks runs --k 59 --queries 50000000 --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 5 --warmup 1 --no-cpu-baseline --parity-sample 200000 > $out/human_runs_k59.json 2> $out/human_runs_k59.err || { tail -3 $out/human_runs_k59.err; exit 1; }
echo "human run blocks k=59 $(python -c "import json;d=json.load(open('$out/human_runs_k59.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity']['mismatches'], d['roofline'].get('kernel'))")"
