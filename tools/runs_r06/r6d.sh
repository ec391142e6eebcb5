#!/bin/bash
# round 6: run blocks behind a sparse table at human scale (>= 4.5e9 q/s in <= 90 GB; round 5: 2.20e9 in 44 GB, 18.7 lines per query),
# k undeclared (depth 23) and declared (depth 31), then the two-tier form of depth 23 on the same index
out=$PWD/gpurun_out/r6d; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));c=d.get('search_counters') or {};print('%.4g q/s  %.3f ms/step' % (d['value'], d['ms_per_step']), 'sparse', d['config'].get('sparse_table_depth'), 'tiers', d['config'].get('sparse_table_tiers'), 'index %.1f GB' % (d['config']['index_bytes']/1e9), 'lines/query', c.get('lines_per_query'), 'parity', d.get('parity'))" $1; }
common="--blocks runs --queries 100000000 --no-c4 --no-c5 --no-sorted --no-live-pmc --no-cpu-baseline --no-variants --counters --parity-sample 2000000 --steps 10 --warmup 2"
MSBWT_VERBOSE=1 timeout -k 10 400 python bench.py $common --query-length-hint 0 --extras-file $out/runs_k_unknown.json > $out/runs_k_unknown.line 2> $out/runs_k_unknown.err || { tail -5 $out/runs_k_unknown.err; exit 1; }
echo "runs, k undeclared: $(show $out/runs_k_unknown.json)"
MSBWT_VERBOSE=1 timeout -k 10 400 python bench.py $common --extras-file $out/runs_k31.json > $out/runs_k31.line 2> $out/runs_k31.err || { tail -5 $out/runs_k31.err; exit 1; }
echo "runs, k = 31 declared: $(show $out/runs_k31.json)"
MSBWT_VERBOSE=1 timeout -k 10 400 python bench.py $common --k 21 --extras-file $out/runs_k21.json > $out/runs_k21.line 2> $out/runs_k21.err || { tail -5 $out/runs_k21.err; exit 1; }
echo "runs, k = 21 declared: $(show $out/runs_k21.json)"
grep -h "sparse table\|load:" $out/runs_k_unknown.err | head -12
