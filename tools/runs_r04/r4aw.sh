#!/bin/bash
# c4r's two modes: which array's rebuild flips it?  one instance; the table alone rebuilt six times, then pair blocks + table four times
out=$PWD/gpurun_out/r4aw; mkdir -p $out
MSBWT_VERBOSE=1 timeout -k 10 220 python tools/rebuild_probe.py c4r 6 4 > $out/rebuild.log 2> $out/rebuild.err || { tail -5 $out/rebuild.err; exit 1; }
cat $out/rebuild.log
