#!/bin/bash
# c4r's two modes follow the SMALL arrays (r4ao), and not the superblock table's pages (r4ap): which one?  pads in front of one array at a time
out=$PWD/gpurun_out/r4aq; mkdir -p $out
for p in 1 2 3; do
  MSBWT_VERBOSE=1 timeout -k 10 500 python tools/placement_probe.py c4r 0 p:0,0,4,0 p:0,0,8,0 p:0,0,16,0 p:0,4,0,0 p:4,0,0,0 p:0,0,0,4 p:0,0,0,8 0 > $out/probe_$p.log 2> $out/probe_$p.err || { tail -5 $out/probe_$p.err; exit 1; }
  echo "process $p"; cut -c1-30,62-84 $out/probe_$p.log; grep "load: blocks\|ticket counters\|status block" $out/probe_$p.err | sed 's/.*pair super/  super/; s/table 0x[0-9a-f]* //; s/filter.*//; s/.msbwt. launch slot: /  /; s/.msbwt. status/  status/'
done
