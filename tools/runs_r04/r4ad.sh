#!/bin/bash
# which of the round's additions to the lanes kernel cost C3 fused its 3 %: variants with the escape step / the counters compiled out
out=$PWD/gpurun_out/r4ad; mkdir -p $out
cp rust-msbwt_amd/libmsbwt_hip.so /tmp/lib_default.so
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" $1; }
for rep in 1 2; do for v in default ab_noesc ab_nocnt ab_none; do
  if [ $v = default ]; then cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so; else cp tools/_variants/$v.so rust-msbwt_amd/libmsbwt_hip.so; fi
  timeout -k 10 300 python bench.py --workload c3 --fused --no-oracle --steps 10 --warmup 2 > $out/c3f_${v}_$rep.json 2> $out/c3f_${v}_$rep.err || exit 1
  echo "c3 fused $v rep$rep $(line $out/c3f_${v}_$rep.json)"
  timeout -k 10 300 python bench.py --workload c2 --no-oracle --steps 20 --warmup 3 > $out/c2_${v}_$rep.json 2> $out/c2_${v}_$rep.err || exit 1
  echo "c2 $v rep$rep $(line $out/c2_${v}_$rep.json)"
done; done
cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so
