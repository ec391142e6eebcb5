#!/bin/bash
# same box: the lanes kernel of commit 97fed54 (escape queries fetch their side-array entry as a search step) against this one
# (fetched and waited for in step A) on the workloads that HAVE escape lines, and on C4 which has none
out=$PWD/gpurun_out/r4ag; mkdir -p $out
cp rust-msbwt_amd/libmsbwt_hip.so /tmp/lib_default.so
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" $1; }
for w in c4r c4x3r c4; do for rep in 1 2; do for v in default old97; do
  if [ $v = default ]; then cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so; else cp tools/_variants/$v.so rust-msbwt_amd/libmsbwt_hip.so; fi
  timeout -k 10 400 python bench.py --workload $w --no-oracle --no-c5 --no-cpu-baseline --steps 10 --warmup 2 > $out/${w}_${v}_$rep.json 2> $out/${w}_${v}_$rep.err || exit 1
  echo "$w $v rep$rep $(line $out/${w}_${v}_$rep.json)"
done; done; done
cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so
