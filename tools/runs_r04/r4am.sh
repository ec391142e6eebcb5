#!/bin/bash
# c4r's two modes: does a launch's time follow where the index sits?  three processes, each building the index six times (plain, plain, behind a
# 40 GiB spacer, plain, behind a 100 GiB spacer, plain)
out=$PWD/gpurun_out/r4am; mkdir -p $out
for p in 1 2 3; do
  MSBWT_VERBOSE=1 timeout -k 10 500 python tools/placement_probe.py c4r > $out/probe_$p.log 2> $out/probe_$p.err || { tail -5 $out/probe_$p.err; exit 1; }
  echo "process $p"; cat $out/probe_$p.log; grep "load: blocks" $out/probe_$p.err
done
