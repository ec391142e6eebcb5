#!/bin/bash
# round 4: the in-library ordering pass -- parity first, then what it costs and brings on C4 / human
out=gpurun_out/r4d; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_two_bit or batch_order or escape or ordered" > $out/parity_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -15 $out/parity_new.log
[ $rc -eq 0 ] || exit 1
for mode in 0 1; do for bits in 24 20 13; do
  [ $mode = 0 ] && [ $bits != 24 ] && continue
  MSBWT_ORDER=$mode MSBWT_ORDER_BITS=$bits timeout -k 10 300 python bench.py --workload c4 --query-kind reads --no-oracle --steps 10 --warmup 2 > $out/c4_order${mode}_b$bits.json 2> $out/c4_order${mode}_b$bits.err || exit 1
  echo "c4 order=$mode bits=$bits $(python -c "import json;d=json.load(open('$out/c4_order${mode}_b$bits.json'));print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])")"
done; done
MSBWT_ORDER=1 timeout -k 10 300 python bench.py --workload c4 --query-kind reads --steps 10 --warmup 2 --no-cpu-baseline --stats-sample 200000 > $out/c4_order1_parity.json 2> $out/c4_order1_parity.err || exit 1
echo "c4 order=1 with oracle: $(python -c "import json;d=json.load(open('$out/c4_order1_parity.json'));print(d['value'], d['parity'])")"
cd /tmp && export TMPDIR=/tmp && MSBWT_ORDER=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_c4_order -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --query-kind reads --no-oracle --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$out/prof_c4_order.json 2> $GRAFT_REPO_ROOT/$out/prof_c4_order.err
cd $GRAFT_REPO_ROOT; find $out/prof_c4_order -name "*kernel_stats.csv" | head -1 | xargs -I{} head -20 {}
