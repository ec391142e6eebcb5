#!/bin/bash
# tile tickets drawn a segment ahead and looked at only then (no wait behind the atomic): parity of the lanes paths, then c4r's modes (three instances
# side by side, per launch) and the set-up-bound lines
out=$PWD/gpurun_out/r4as; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes or packed or escape or order or ticket or large or batch" > $out/parity.log 2>&1 || { tail -20 $out/parity.log; exit 1; }
tail -1 $out/parity.log
timeout -k 10 500 python tools/instance_probe.py c4r 3 2 > $out/inst.log 2> $out/inst.err || { tail -5 $out/inst.err; exit 1; }
grep "side array 1" $out/inst.log
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" $1; }
timeout -k 10 300 python bench.py --workload c3 --fused --no-oracle --steps 10 --warmup 2 > $out/c3f.json 2> $out/c3f.err || exit 1
echo "c3 fused $(line $out/c3f.json)"
timeout -k 10 300 python bench.py --workload c2 --no-oracle --steps 20 --warmup 3 > $out/c2.json 2> $out/c2.err || exit 1
echo "c2 $(line $out/c2.json)"
