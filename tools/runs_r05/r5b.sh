#!/bin/bash
# round 5: the whole GPU suite with the sparse table in (five modes of the parity file), then the human-scale line with the table off / on
out=gpurun_out/r5b; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -25 $out/gputests.log
for mode in auto 0; do
  MSBWT_VERBOSE=1 MSBWT_SPARSE_TABLE=$mode timeout -k 10 500 python bench.py --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 --counters --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/human_sparse_$mode.json 2> $out/human_sparse_$mode.err || { tail -5 $out/human_sparse_$mode.err; exit 1; }
  grep -E "sparse table|load:" $out/human_sparse_$mode.err | tail -12
  echo "human sparse=$mode $(python -c "import json;d=json.load(open('$out/human_sparse_$mode.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity'], json.dumps(d['search_counters']['raw']), d['search_counters']['lines_per_query'], d['config'].get('index_bytes'))")"
done
