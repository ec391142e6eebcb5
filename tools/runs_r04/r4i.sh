#!/bin/bash
# round 4: the ordering pass, third shape (persistent level 0, window-local places, chunk-major unsort)
out=gpurun_out/r4j; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_two_bit or batch_order" > $out/parity_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -5 $out/parity_new.log
[ $rc -eq 0 ] || exit 1
for cfg in "1024 128" "1024 256" "2048 128"; do set -- $cfg
  MSBWT_ORDER=1 MSBWT_ORDER_WG0=$1 MSBWT_ORDER_WG1=$2 timeout -k 10 300 python bench.py --workload c4 --query-kind reads --no-oracle --steps 10 --warmup 2 > $out/c4_wg$1_$2.json 2> $out/c4_wg$1_$2.err || exit 1
  echo "c4 order=1 wg0=$1 wg1=$2 $(python -c "import json;d=json.load(open('$out/c4_wg$1_$2.json'));print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])")"
done
cd /tmp && export TMPDIR=/tmp
MSBWT_ORDER=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --query-kind reads --no-oracle --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$out/prof.json 2> $GRAFT_REPO_ROOT/$out/prof.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob('$out/prof/*kernel_stats.csv')[0])):
    if 'k_order' in r['Name'] or 'k_count' in r['Name']:
        print(r['Name'][:75].ljust(76), r['Calls'], "%.3f ms" % (float(r['AverageNs'])/1e6))
PY
