#!/bin/bash
# NOT RUN in round 4 (the GPU budget was spent): the first call of the next round on the two modes of the C4-sized lines.
#  1. does a plain gather over a 3 GiB table (the pair blocks' size) see two rates, allocation after allocation?
#  2. three instances side by side per process, pair blocks from hipMalloc / 1 GiB chunks / hipDeviceMallocContiguous: which modes come up?
out=$PWD/gpurun_out/next_placement; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_placement tools/ubench_placement.hip > $out/build.log 2>&1 || { tail $out/build.log; exit 1; }
for p in 1 2; do
  timeout -k 10 200 /tmp/ubench_placement 3 0 0 0 0 1 2 4 8 16 32 64 128 0 0 > $out/gather3_$p.log 2>&1 || { tail -5 $out/gather3_$p.log; exit 1; }
  echo "gather over 3 GiB, process $p"; cat $out/gather3_$p.log
done
for mode in malloc chunks contiguous; do for p in 1 2; do
  MSBWT_BIG_ALLOC=$mode MSBWT_VERBOSE=1 timeout -k 10 300 python tools/instance_probe.py c4r 3 1 1 > $out/inst_${mode}_$p.log 2> $out/inst_${mode}_$p.err || { tail -5 $out/inst_${mode}_$p.err; exit 1; }
  echo "pair blocks by $mode, process $p"; cat $out/inst_${mode}_$p.log; grep "physical chunks\|contiguous bytes" $out/inst_${mode}_$p.err | head -3
done; done
