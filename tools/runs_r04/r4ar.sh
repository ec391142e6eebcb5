#!/bin/bash
# c4r's two modes: a property of the index INSTANCE?  three instances side by side, timed in turn; then without the side array
out=$PWD/gpurun_out/r4ar; mkdir -p $out
for p in 1 2; do
  timeout -k 10 500 python tools/instance_probe.py c4r 3 3 > $out/inst_$p.log 2> $out/inst_$p.err || { tail -5 $out/inst_$p.err; exit 1; }
  echo "process $p"; cat $out/inst_$p.log
done
