#!/bin/bash
# does the rate of random line fetches from a 68 GiB table follow where the table sits (tools/ubench_placement.hip)?  two processes
out=$PWD/gpurun_out/r4an; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_placement tools/ubench_placement.hip > $out/build.log 2>&1 || { tail $out/build.log; exit 1; }
for p in 1 2; do
  timeout -k 10 300 /tmp/ubench_placement 68 > $out/placement_$p.log 2>&1 || { tail -5 $out/placement_$p.log; exit 1; }
  echo "process $p"; cat $out/placement_$p.log
done
