#!/bin/bash
# round 6, first call after merging r6-deep-sparse-table: the whole GPU suite, smoke(), a soak over three seeds (sparse depths 25 / 27 / 28 among the settings)
out=gpurun_out/r6a; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -4 $out/gputests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
for seed in 111 112 113; do
  STRESS_SEED=$seed timeout -k 10 260 python tools/stress_parity.py 150 > $out/soak_seed$seed.log 2>&1; rc=$?
  echo "seed $seed rc=$rc: $(tail -1 $out/soak_seed$seed.log)"
  [ $rc -eq 0 ] || { tail -20 $out/soak_seed$seed.log; exit 1; }
done
# depth 29 has never run on a GPU: its counts test, then the metric's line with it
timeout -k 10 200 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu -k "counts_with and 29" 2>&1 | tail -2
timeout -k 10 400 python bench.py --sparse-depth 29 --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --counters --parity-sample 2000000 > $out/human_depth29.json 2> $out/human_depth29.log
python -c "import json;r=json.loads(open('$out/human_depth29.json').read().strip().splitlines()[-1]);print('depth 29:', r['value'], r['ms_per_step'], r['search_counters']['lines_per_query'], r['parity'], r['config']['index_bytes'])"
# depths 30 / 31 (the xwide layout) have never run on a GPU either
timeout -k 10 300 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu -k "30 or 31" 2>&1 | tail -2
for d in 31 -2; do   # explicit depth 31, then the default (bench declares its k: the automatic depth should come out as 31)
  extra=""; [ $d -gt 0 ] && extra="--sparse-depth $d"
  timeout -k 10 400 python bench.py $extra --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --counters --parity-sample 2000000 > $out/human_depth$d.json 2> $out/human_depth$d.log
  python -c "import json;r=json.loads(open('$out/human_depth$d.json').read().strip().splitlines()[-1]);print('sparse depth', r['config']['sparse_table_depth'], r['value'], r['ms_per_step'], r['search_counters']['lines_per_query'], r['parity'], r['config']['index_bytes'])"
done
