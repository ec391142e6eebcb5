#!/bin/bash
# c4r's two modes (18 / 20.5 ms, per process): six runs with the addresses of every array logged (MSBWT_VERBOSE), to see what the mode follows
out=$PWD/gpurun_out/r4al; mkdir -p $out
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" $1; }
for rep in 1 2 3 4 5 6; do
  MSBWT_VERBOSE=1 timeout -k 10 400 python bench.py --workload c4r --no-oracle --no-c5 --no-cpu-baseline --steps 10 --warmup 2 > $out/c4r_$rep.json 2> $out/c4r_$rep.err || exit 1
  echo "c4r rep$rep $(line $out/c4r_$rep.json)"
  grep "load: blocks\|buffers:" $out/c4r_$rep.err | sort -u | head -4
done
