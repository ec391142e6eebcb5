#!/bin/bash
# round 5 experiment: non-temporal LDS-DMA loads for the index lines (aux = nt) against the default, same box: human scale, C4, C3 fused
out=gpurun_out/r5v; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for rep in 1 2; do
bash tools/sweep_variants.sh "--no-c4 --no-live-pmc --no-sorted" default tools/_variants/dma_nt.so 2>&1 | grep "q/s" | sed "s/^/human rep$rep /"
bash tools/sweep_variants.sh "--workload c4 --query-kind reads" default tools/_variants/dma_nt.so 2>&1 | grep "q/s" | sed "s/^/c4 rep$rep /"
bash tools/sweep_variants.sh "--workload c3 --fused" default tools/_variants/dma_nt.so 2>&1 | grep "q/s" | sed "s/^/c3f rep$rep /"
done
