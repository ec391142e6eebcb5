#!/bin/bash
# round 4: does the library's batch ordering pay?  same box, pass off / on: C4, C4 with repeats, C3 (matrix), human scale
out=gpurun_out/r4k; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
run() { # name mode args...
  name=$1; mode=$2; shift 2
  MSBWT_ORDER=$mode MSBWT_ORDER_WG1=256 timeout -k 10 500 python bench.py "$@" --no-oracle --steps 10 --warmup 2 > $out/${name}_order$mode.json 2> $out/${name}_order$mode.err || return 1
  echo "$name order=$mode $(python -c "import json;d=json.load(open('$out/${name}_order$mode.json'));print('%.4g q/s  %.3f ms/step  kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))")"
}
for mode in 0 1; do run c4 $mode --workload c4 --query-kind reads || exit 1; done
for mode in 0 1; do run c4r $mode --workload c4r || exit 1; done
for mode in 0 1; do run c3 $mode --workload c3 || exit 1; done
for mode in 0 1; do run c4_random $mode --workload c4 || exit 1; done
for mode in 0 1; do run c4_half $mode --workload c4 --query-kind reads --queries 30000000 || exit 1; done
for mode in 0 1; do run human $mode --no-c5 --no-c4 --no-live-pmc --no-sorted || exit 1; done
