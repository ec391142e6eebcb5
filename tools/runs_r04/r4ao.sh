#!/bin/bash
# c4r's two modes: is it where the SMALL arrays sit (superblock table, ticket counters)?  three processes, each building the index nine times with small
# pads in front (the big arrays stay where they are)
out=$PWD/gpurun_out/r4ao; mkdir -p $out
for p in 1 2 3; do
  MSBWT_VERBOSE=1 timeout -k 10 500 python tools/placement_probe.py c4r 0 -4 -8 -16 -64 -256 -1024 -4096 0 > $out/probe_$p.log 2> $out/probe_$p.err || { tail -5 $out/probe_$p.err; exit 1; }
  echo "process $p"; paste -d' ' <(cut -c1-75 $out/probe_$p.log) <(grep "load: blocks" $out/probe_$p.err | sed 's/.*pair super/super/; s/side.*//')
done
