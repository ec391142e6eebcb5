#!/bin/bash
# do the search counters cost the fused read-window kernel anything?  same box: default (no counters in the kReads instantiations) vs the variant that keeps them
out=gpurun_out/r4z; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for rep in 1 2; do
tools/sweep_variants.sh "--workload c3 --fused --no-live-pmc --no-sorted" default tools/_variants/cnt_reads.so 2>&1 | grep -E "q/s|rror" | tee -a $out/c3_fused.log
done
tools/sweep_variants.sh "--workload c2 --no-live-pmc --no-sorted" default 2>&1 | grep -E "q/s|rror" | tee -a $out/c2.log
