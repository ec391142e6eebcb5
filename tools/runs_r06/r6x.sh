#!/bin/bash
# round 6: the two-tier table on the largest real MSBWT of reads WITH errors this box builds (c4x3: 3.87e7 reads, 5.84e9 symbols, suffix-sorted on the
# host in ~3 min): complete depth-23 table / two-tier / no sparse table, 1e8 read-derived 31-mers, k undeclared, parity on 2e6 each
out=$PWD/gpurun_out/r6x; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
( while sleep 60; do echo "... $(date +%T)"; done ) & hb=$!
trap "kill $hb" EXIT
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));c=d.get('search_counters') or {};print('%.4g q/s  %.3f ms/step' % (d['value'], d['ms_per_step']), 'sparse', d['config'].get('sparse_table_depth'), 'tiers', d['config'].get('sparse_table_tiers'), 'table %.2f GB' % ((d['config']['sparse_table']['bytes'])/1e9), 'entries', d['config']['sparse_table']['entries'], 'filtered', d['config']['sparse_table']['filtered'], 'index %.1f GB' % (d['config']['index_bytes']/1e9), 'lines/query', c.get('lines_per_query'), 'parity', d.get('parity'))" $1; }
common="--workload c4x3 --query-kind reads --query-length-hint 0 --no-variants --no-cpu-baseline --counters --parity-sample 2000000 --steps 10 --warmup 2"
MSBWT_VERBOSE=1 timeout -k 10 900 python bench.py $common --extras-file $out/c4x3_complete.json > $out/c4x3_complete.line 2> $out/c4x3_complete.err || { tail -5 $out/c4x3_complete.err; exit 1; }
echo "c4x3 complete: $(show $out/c4x3_complete.json)"
MSBWT_VERBOSE=1 timeout -k 10 400 python bench.py $common --sparse-tiers 1 --extras-file $out/c4x3_two_tier.json > $out/c4x3_two_tier.line 2> $out/c4x3_two_tier.err || { tail -5 $out/c4x3_two_tier.err; exit 1; }
echo "c4x3 two-tier: $(show $out/c4x3_two_tier.json)"
MSBWT_VERBOSE=1 timeout -k 10 400 python bench.py $common --sparse-depth 0 --extras-file $out/c4x3_sparse_off.json > $out/c4x3_sparse_off.line 2> $out/c4x3_sparse_off.err || { tail -5 $out/c4x3_sparse_off.err; exit 1; }
echo "c4x3 no sparse table: $(show $out/c4x3_sparse_off.json)"
grep -h "sparse table" $out/c4x3_two_tier.err | head -4
