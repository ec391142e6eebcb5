#!/bin/bash
# round 5: repeats at three times C4's size (c4x3r: 5.84e9 symbols, copy numbers x 3; 169 s of host suffix sorting) with the sparse table, and without
out=gpurun_out/r5t; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for mode in auto 0; do
MSBWT_VERBOSE=1 MSBWT_SPARSE_TABLE=$mode timeout -k 10 900 python bench.py --workload c4x3r --steps 5 --warmup 1 --counters --no-cpu-baseline --stats-sample 200000 --parity-sample 1000000 > $out/c4x3r_$mode.json 2> $out/c4x3r_$mode.err || { tail -5 $out/c4x3r_$mode.err; exit 1; }
grep -E "sparse table:" $out/c4x3r_$mode.err | tail -2
echo "c4x3r sparse=$mode $(python -c "import json;d=json.load(open('$out/c4x3r_$mode.json'));c=d['search_counters'];print(d['value'], d['roofline']['kernel_ms'], d['parity'], c['lines_per_query'], c['second_line_rate'], c['escape_query_fraction'], d['config']['index_bytes'])")"
done
