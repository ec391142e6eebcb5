#!/bin/bash
# after the packed-query split of the lanes kernel (kPacked): targeted parity, then the same-box A/B against the round-3 tree
out=$PWD/gpurun_out/r4ac; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
(cd .r03_tree && python -c "import __graft_entry__ as g; g.build()" > $out/build_r03.log 2>&1) || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed or escape or order or run or single or inline or small" > $out/parity.log 2>&1 || { tail -20 $out/parity.log; exit 1; }
tail -2 $out/parity.log
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));c5=d.get('c5_random_1e9');print('%.4g q/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']), '' if not c5 else 'c5 %.4g q/s %.2f ms' % (c5['value'], c5['ms_per_pass']))" $1; }
for rep in 1 2; do for tree in r03 r04; do
  dir=$PWD; [ $tree = r03 ] && dir=$PWD/.r03_tree
  (cd $dir && timeout -k 10 300 python bench.py --workload c2 --no-oracle --steps 20 --warmup 3 > $out/c2_${tree}_$rep.json 2> $out/c2_${tree}_$rep.err) || exit 1
  echo "c2 $tree rep$rep $(line $out/c2_${tree}_$rep.json)"
  (cd $dir && timeout -k 10 300 python bench.py --workload c3 --fused --no-oracle --steps 10 --warmup 2 > $out/c3f_${tree}_$rep.json 2> $out/c3f_${tree}_$rep.err) || exit 1
  echo "c3 fused $tree rep$rep $(line $out/c3f_${tree}_$rep.json)"
done; done
for tree in r03 r04; do
  dir=$PWD; [ $tree = r03 ] && dir=$PWD/.r03_tree
  (cd $dir && timeout -k 10 500 python bench.py --no-oracle --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 > $out/human_${tree}.json 2> $out/human_${tree}.err) || exit 1
  echo "human $tree $(line $out/human_${tree}.json)"
done
