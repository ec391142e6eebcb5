#!/bin/bash
# round 6: does the non-temporal line hint pay on a C4-sized index whose LOOKUP table (depth 31, 6 GB) is far beyond the caches while its pair blocks (2.6 GB) are not?
out=$PWD/gpurun_out/r6v; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
show() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']), 'sparse', d['config'].get('sparse_table_depth'))" $1; }
for rep in 1 2; do for mode in auto 1; do
  for wl in c4 c4r; do
    MSBWT_STREAM_LINES=$mode timeout -k 10 300 python bench.py --workload $wl --query-kind reads --no-variants --no-oracle --steps 20 --warmup 3 --extras-file $out/${wl}_${mode}_$rep.json > $out/${wl}_${mode}_$rep.line 2> $out/${wl}_${mode}_$rep.err || exit 1
    echo "$wl declared, streaming $mode rep$rep: $(show $out/${wl}_${mode}_$rep.json)"
  done
  MSBWT_STREAM_LINES=$mode timeout -k 10 300 python bench.py --workload c3 --fused --no-variants --no-oracle --steps 10 --warmup 2 --extras-file $out/c3f_${mode}_$rep.json > $out/c3f_${mode}_$rep.line 2> $out/c3f_${mode}_$rep.err || exit 1
  echo "c3 fused declared, streaming $mode rep$rep: $(show $out/c3f_${mode}_$rep.json)"
done; done
