#!/bin/bash
# round 6: what is left of the A/B gap on the small indexes -- the deep direct table kept beside the sparse one (73 GB more in HBM), or the kernel?
# C3 fused and C2 read-derived, k undeclared: round 5's tree / this tree / this tree with the direct table as round 5 built it (flat 13 -> packed 15)
out=$PWD/gpurun_out/r6h; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
(cd .r05_tree && python -c "import __graft_entry__ as g; g.build()" > $out/build_r05.log 2>&1) || { tail -5 $out/build_r05.log; exit 1; }
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']), d['config'].get('sparse_table_depth'), d['config'].get('direct_table_depth'), d['config'].get('index_bytes'))" $1; }
for rep in 1 2; do
  (cd .r05_tree && timeout -k 10 300 python bench.py --workload c3 --fused --no-oracle --steps 10 --warmup 2 > $out/c3f_r05_$rep.json 2> $out/c3f_r05_$rep.err) || exit 1
  echo "c3 fused r05 rep$rep $(line $out/c3f_r05_$rep.json)"
  timeout -k 10 300 python bench.py --workload c3 --fused --query-length-hint 0 --no-variants --no-oracle --steps 10 --warmup 2 > $out/c3f_r06_$rep.json 2> $out/c3f_r06_$rep.err || exit 1
  echo "c3 fused r06 rep$rep $(line $out/c3f_r06_$rep.json)"
  MSBWT_TABLE_DEPTH=13 MSBWT_TABLE_PACKED=1 timeout -k 10 300 python bench.py --workload c3 --fused --query-length-hint 0 --no-variants --no-oracle --steps 10 --warmup 2 > $out/c3f_r06s_$rep.json 2> $out/c3f_r06s_$rep.err || exit 1
  echo "c3 fused r06, shallow direct table rep$rep $(line $out/c3f_r06s_$rep.json)"
  (cd .r05_tree && timeout -k 10 300 python bench.py --workload c2 --query-kind reads --no-oracle --steps 20 --warmup 3 > $out/c2_r05_$rep.json 2> $out/c2_r05_$rep.err) || exit 1
  echo "c2 reads r05 rep$rep $(line $out/c2_r05_$rep.json)"
  timeout -k 10 300 python bench.py --workload c2 --query-kind reads --query-length-hint 0 --no-variants --no-oracle --steps 20 --warmup 3 > $out/c2_r06_$rep.json 2> $out/c2_r06_$rep.err || exit 1
  echo "c2 reads r06 rep$rep $(line $out/c2_r06_$rep.json)"
  MSBWT_TABLE_DEPTH=13 MSBWT_TABLE_PACKED=1 timeout -k 10 300 python bench.py --workload c2 --query-kind reads --query-length-hint 0 --no-variants --no-oracle --steps 20 --warmup 3 > $out/c2_r06s_$rep.json 2> $out/c2_r06s_$rep.err || exit 1
  echo "c2 reads r06, shallow direct table rep$rep $(line $out/c2_r06s_$rep.json)"
done
