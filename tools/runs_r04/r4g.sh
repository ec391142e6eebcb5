#!/bin/bash
# what would the ordered search cost if its counts did not have to be scattered?  (MSBWT_PLACED_STORE=9: counts stay in sorted order -- wrong places, timing only)
out=gpurun_out/r4g; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for store in 9 0; do
MSBWT_ORDER=1 MSBWT_PLACED_STORE=$store timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_store$store -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --query-kind reads --no-oracle --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$out/prof_store$store.json 2> $GRAFT_REPO_ROOT/$out/prof_store$store.err
echo "store=$store"; find $GRAFT_REPO_ROOT/$out/prof_store$store -name "*kernel_stats.csv" | head -1 | xargs -I{} awk -F'","' '/k_order|k_count/ {print substr($1,1,70), $2, $4}' {}
done
