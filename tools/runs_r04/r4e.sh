#!/bin/bash
# round 4: the restructured ordering pass (10 + 12 bits, unrolled) and how the placed counts are stored
out=gpurun_out/r4e; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_two_bit or batch_order" > $out/parity_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -5 $out/parity_new.log
[ $rc -eq 0 ] || exit 1
for store in 0 1 2; do for bits in 22 20 24; do
  [ $store != 0 ] && [ $bits != 22 ] && continue
  MSBWT_ORDER=1 MSBWT_ORDER_BITS=$bits MSBWT_PLACED_STORE=$store timeout -k 10 300 python bench.py --workload c4 --query-kind reads --no-oracle --steps 10 --warmup 2 > $out/c4_store${store}_b$bits.json 2> $out/c4_store${store}_b$bits.err || exit 1
  echo "c4 order=1 store=$store bits=$bits $(python -c "import json;d=json.load(open('$out/c4_store${store}_b$bits.json'));print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])")"
done; done
cd /tmp && export TMPDIR=/tmp && MSBWT_ORDER=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_c4_order -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --query-kind reads --no-oracle --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$out/prof_c4_order.json 2> $GRAFT_REPO_ROOT/$out/prof_c4_order.err
cd $GRAFT_REPO_ROOT; find $out/prof_c4_order -name "*kernel_stats.csv" | head -1 | xargs -I{} grep -E "k_order|k_count" {} | cut -c1-60,200-
find $out/prof_c4_order -name "*kernel_stats.csv" | head -1 | xargs -I{} awk -F'","' '/k_order|k_count/ {print $1, $2, $4}' {}
