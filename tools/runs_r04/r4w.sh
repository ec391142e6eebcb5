#!/bin/bash
# round 4: the ordering passes with LDS-staged runs -- parity, then pass off / on on one box
out=gpurun_out/r4w; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_two_bit or batch_order" > $out/parity_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -5 $out/parity_new.log
[ $rc -eq 0 ] || exit 1
run() { name=$1; mode=$2; shift 2
  MSBWT_ORDER=$mode timeout -k 10 500 python bench.py "$@" --no-oracle --steps 10 --warmup 2 > $out/${name}_order$mode.json 2> $out/${name}_order$mode.err || return 1
  echo "$name order=$mode $(python -c "import json;d=json.load(open('$out/${name}_order$mode.json'));print('%.4g q/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")"; }
for mode in 0 1; do run c4 $mode --workload c4 --query-kind reads || exit 1; done
for mode in 0 1; do run c4r $mode --workload c4r || exit 1; done
for mode in 0 1; do run c3 $mode --workload c3 || exit 1; done
cd /tmp && export TMPDIR=/tmp
MSBWT_ORDER=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --query-kind reads --no-oracle --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$out/prof.json 2> $GRAFT_REPO_ROOT/$out/prof.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,re
for r in csv.DictReader(open(glob.glob('$out/prof/*kernel_stats.csv')[0])):
    if 'k_order' in r['Name'] or 'k_count' in r['Name']:
        print("  %-24s %s calls  %.3f ms" % (re.search(r'(k_[a-z_]+)', r['Name']).group(1), r['Calls'], float(r['AverageNs'])/1e6))
PY
