#!/bin/bash
# superblock table spread to one line per 4 KiB page: do c4r's two modes collapse?  pair-index parity first, then three processes of the placement probe
out=$PWD/gpurun_out/r4ap; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair or superblock or clone or budget" > $out/parity.log 2>&1 || { tail -20 $out/parity.log; exit 1; }
tail -1 $out/parity.log
for p in 1 2 3; do
  MSBWT_VERBOSE=1 timeout -k 10 500 python tools/placement_probe.py c4r 0 -4 -8 -16 -64 0 > $out/probe_$p.log 2> $out/probe_$p.err || { tail -5 $out/probe_$p.err; exit 1; }
  echo "process $p"; paste -d' ' <(cut -c1-75 $out/probe_$p.log) <(grep "load: blocks" $out/probe_$p.err | sed 's/.*pair super/super/; s/table.*//')
done
