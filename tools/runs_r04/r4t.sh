#!/bin/bash
# one-off: the repeat-bearing genome at three times C4's size (5.8e9 symbols, copy numbers x3): escape lines with and without the side array
out=gpurun_out/r4t; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for side in on off; do
  extra=""; [ $side = off ] && extra="--no-table-side"
  timeout -k 10 900 python bench.py --workload c4x3r --steps 10 --warmup 2 --counters --no-cpu-baseline --stats-sample 200000 --parity-sample 500000 $extra > $out/c4x3r_side_$side.json 2> $out/c4x3r_side_$side.err || exit 1
  echo "c4x3r side=$side $(python -c "import json;d=json.load(open('$out/c4x3r_side_$side.json'));c=d['search_counters'];print('%.4g q/s %.2f ms' % (d['value'], d['ms_per_step']), d['parity'], 'lines/query %.3f second %.3f escape queries %.4f restarts %.4f' % (c['lines_per_query'], c['second_line_rate'], c['escape_query_fraction'], c['escape_restart_fraction']), c['table'])")"
done
