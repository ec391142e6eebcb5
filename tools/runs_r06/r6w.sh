#!/bin/bash
# round 6: un-profiled bench lines of BASELINE's other configurations on the final tree, with the oracle's parity sample, k declared and undeclared
out=$PWD/gpurun_out/r6w; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']), 'frac', d['roofline'].get('frac'), 'sparse', d['config'].get('sparse_table_depth'), 'parity', d['parity'])" $1; }
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --no-variants --extras-file $out/$name.json > $out/$name.line 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }; echo "$name: $(show $out/$name.json)"; }
run c2_declared --workload c2
run c2_k_unknown --workload c2 --query-length-hint 0
run c2_reads_declared --workload c2 --query-kind reads
run c3_fused_declared --workload c3 --fused
run c3_fused_k_unknown --workload c3 --fused --query-length-hint 0
run c4_random_declared --workload c4
run c4_random_k_unknown --workload c4 --query-length-hint 0
run c4_reads_k_unknown --workload c4 --query-kind reads --query-length-hint 0
run c4_reads_two_tier --workload c4 --query-kind reads --query-length-hint 0 --sparse-tiers 1
