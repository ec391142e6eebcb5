#!/bin/bash
# round 6: the two-tier KERNEL at human scale -- the error-free index with the two-tier form forced (nothing occurs once there, so the table holds every suffix at
# the two-tier load: 60 GB, the size the estimate for a read set WITH errors gives): what the solid queries of such a set would run at
out=$PWD/gpurun_out/r6k; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
MSBWT_VERBOSE=1 timeout -k 10 400 python bench.py --query-length-hint 0 --sparse-tiers 1 --no-c4 --no-c5 --no-sorted --no-variants --no-live-pmc --no-cpu-baseline --counters --parity-sample 2000000 --steps 10 --warmup 2 --extras-file $out/human_two_tier.json > $out/human_two_tier.line 2> $out/human_two_tier.err || { tail -5 $out/human_two_tier.err; exit 1; }
python -c "import json;d=json.load(open('$out/human_two_tier.json'));c=d['search_counters'];print('%.4g q/s  %.3f ms/step kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']), 'sparse', d['config']['sparse_table_depth'], 'tiers', d['config']['sparse_table_tiers'], 'table %.1f GB' % (d['config']['sparse_table']['bytes']/1e9), 'index %.1f GB' % (d['config']['index_bytes']/1e9), 'lines/query', c['lines_per_query'], 'fallbacks', c['raw'].get('tier_fallbacks'), 'parity', d['parity'])"
grep -h "sparse table" $out/human_two_tier.err | head -3
