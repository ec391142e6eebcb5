#!/bin/bash
# round 5 experiment: which cache-policy bits for the LDS-DMA loads of the index lines (aux: 1 = sc0, 2 = nt, 16 = sc1 and sums) at human scale, and nt at c4x3r's size
out=gpurun_out/r5w; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
bash tools/sweep_variants.sh "--no-c4 --no-live-pmc --no-sorted" default tools/_variants/dma_nt.so tools/_variants/dma_aux1.so tools/_variants/dma_aux3.so tools/_variants/dma_aux16.so tools/_variants/dma_aux18.so tools/_variants/dma_aux19.so default 2>&1 | grep "q/s" | sed "s/^/human /"
bash tools/sweep_variants.sh "--workload c4x3r" default tools/_variants/dma_nt.so 2>&1 | grep "q/s" | sed "s/^/c4x3r /"
