#!/bin/bash
# escape entries fetched (and waited for) in step A, no per-step state: parity of the escape paths, then C3 fused / C2 against the variants without
out=$PWD/gpurun_out/r4af; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "escape or side or packed or order or counters" > $out/parity.log 2>&1 || { tail -20 $out/parity.log; exit 1; }
tail -2 $out/parity.log
cp rust-msbwt_amd/libmsbwt_hip.so /tmp/lib_default.so
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" $1; }
for rep in 1 2; do for v in default ab_nocnt ab_none; do
  if [ $v = default ]; then cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so; else cp tools/_variants/$v.so rust-msbwt_amd/libmsbwt_hip.so; fi
  timeout -k 10 300 python bench.py --workload c3 --fused --no-oracle --steps 10 --warmup 2 > $out/c3f_${v}_$rep.json 2> $out/c3f_${v}_$rep.err || exit 1
  echo "c3 fused $v rep$rep $(line $out/c3f_${v}_$rep.json)"
  timeout -k 10 300 python bench.py --workload c2 --no-oracle --steps 20 --warmup 3 > $out/c2_${v}_$rep.json 2> $out/c2_${v}_$rep.err || exit 1
  echo "c2 $v rep$rep $(line $out/c2_${v}_$rep.json)"
done; done
cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so
timeout -k 10 400 python bench.py --workload c4r --no-oracle --no-c5 --no-cpu-baseline --steps 10 --warmup 2 > $out/c4r.json 2> $out/c4r.err || exit 1
echo "c4r $(line $out/c4r.json)"
timeout -k 10 400 python bench.py --workload c4x3r --no-oracle --no-c5 --no-cpu-baseline --steps 10 --warmup 2 > $out/c4x3r.json 2> $out/c4x3r.err || exit 1
echo "c4x3r $(line $out/c4x3r.json)"
