#!/bin/bash
# same box: the round-4 tree (.r04_tree) against this one WITH THE SPARSE TABLE OFF (the direct-table kernels of both rounds side by side: did round 5's
# additions cost the old paths anything?) and against this one as it ships -- C2 random 21-mers, C3 fused, human scale + the 1e9-random-31-mer line
out=$PWD/gpurun_out/r5ab; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
(cd .r04_tree && python -c "import __graft_entry__ as g; g.build()" > $out/build_r04.log 2>&1) || { tail -5 $out/build_r04.log; exit 1; }
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));c5=d.get('c5_random_1e9');print('%.4g q/s  %.3f ms/step' % (d['value'], d['ms_per_step']), '' if not c5 else 'c5 %.4g q/s %.2f ms' % (c5['value'], c5['ms_per_pass']))" $1; }
for rep in 1 2; do for tree in r04 r05off r05; do
  dir=$PWD; [ $tree = r04 ] && dir=$PWD/.r04_tree
  sp=auto; [ $tree = r05off ] && sp=0
  (cd $dir && MSBWT_SPARSE_TABLE=$sp timeout -k 10 300 python bench.py --workload c2 --no-oracle --steps 20 --warmup 3 > $out/c2_${tree}_$rep.json 2> $out/c2_${tree}_$rep.err) || exit 1
  echo "c2 $tree rep$rep $(line $out/c2_${tree}_$rep.json)"
  (cd $dir && MSBWT_SPARSE_TABLE=$sp timeout -k 10 300 python bench.py --workload c3 --fused --no-oracle --steps 10 --warmup 2 > $out/c3f_${tree}_$rep.json 2> $out/c3f_${tree}_$rep.err) || exit 1
  echo "c3 fused $tree rep$rep $(line $out/c3f_${tree}_$rep.json)"
done; done
for tree in r04 r05off r05; do
  dir=$PWD; [ $tree = r04 ] && dir=$PWD/.r04_tree
  sp=auto; [ $tree = r05off ] && sp=0
  (cd $dir && MSBWT_SPARSE_TABLE=$sp timeout -k 10 500 python bench.py --no-oracle --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 > $out/human_${tree}.json 2> $out/human_${tree}.err) || exit 1
  echo "human $tree $(line $out/human_${tree}.json)"
done
