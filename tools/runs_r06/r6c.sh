#!/bin/bash
# round 6: same box, round 5's final tree (.r05_tree = main at the end of round 5) against the merged one -- the lines that do NOT use the wide layout must
# not have lost anything to it (C2 random / read-derived 21-mers, C3 fused, the 1e9-random-31-mer line), and the metric's line with k unknown (depth 23)
# and declared (depth 27)
out=$PWD/gpurun_out/r6c; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
(cd .r05_tree && python -c "import __graft_entry__ as g; g.build()" > $out/build_r05.log 2>&1) || { tail -5 $out/build_r05.log; exit 1; }
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));c5=(d.get('c5_random_1e9') or {}).get('value') or (d.get('extras') or {}).get('c5_random_1e9_qps');print('%.4g q/s  %.3f ms/step' % (d['value'], d['ms_per_step']), '' if not c5 else 'c5 %.4g q/s' % c5, d['config'].get('sparse_table_depth'))" $1; }
for rep in 1 2; do for tree in r05 r06; do
  dir=$PWD; [ $tree = r05 ] && dir=$PWD/.r05_tree
  hint=""; [ $tree = r06 ] && hint="--query-length-hint 0 --no-variants"   # (k undeclared: the same tables as round 5's tree builds)
  for kind in random reads; do
    (cd $dir && timeout -k 10 300 python bench.py --workload c2 --query-kind $kind $hint --no-oracle --steps 20 --warmup 3 > $out/c2_${kind}_${tree}_$rep.json 2> $out/c2_${kind}_${tree}_$rep.err) || exit 1
    echo "c2 $kind $tree rep$rep $(line $out/c2_${kind}_${tree}_$rep.json)"
  done
  (cd $dir && timeout -k 10 300 python bench.py --workload c3 --fused $hint --no-oracle --steps 10 --warmup 2 > $out/c3f_${tree}_$rep.json 2> $out/c3f_${tree}_$rep.err) || exit 1
  echo "c3 fused $tree rep$rep $(line $out/c3f_${tree}_$rep.json)"
done; done
(cd .r05_tree && timeout -k 10 500 python bench.py --no-oracle --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 > $out/human_r05.json 2> $out/human_r05.err) || exit 1
echo "human r05 $(line $out/human_r05.json)"
timeout -k 10 500 python bench.py --query-length-hint 0 --no-oracle --no-c4 --no-live-pmc --no-sorted --no-variants --steps 10 --warmup 2 > $out/human_r06_k_unknown.json 2> $out/human_r06_k_unknown.err || exit 1
echo "human r06, k unknown $(line $out/human_r06_k_unknown.json)"
timeout -k 10 500 python bench.py --no-oracle --no-c4 --no-live-pmc --no-sorted --no-variants --steps 10 --warmup 2 > $out/human_r06.json 2> $out/human_r06.err || exit 1
echo "human r06, k declared $(line $out/human_r06.json)"
