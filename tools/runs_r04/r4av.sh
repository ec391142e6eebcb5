#!/bin/bash
# c4r's two modes: do they follow where the KERNEL ARGUMENTS live (HIP_FORCE_DEV_KERNARG = 0: host memory, 1: device memory)?  three instances each
out=$PWD/gpurun_out/r4av; mkdir -p $out
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python tools/instance_probe.py c4r 3 1 1 > $out/inst_$v.log 2> $out/inst_$v.err || { tail -5 $out/inst_$v.err; exit 1; }
  echo "HIP_FORCE_DEV_KERNARG=$v"; cat $out/inst_$v.log
done
