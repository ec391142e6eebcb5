#!/bin/bash
# round 5 experiment, careful: default against nt (aux 2) and aux 3 at human scale, alternating, four rounds on one box
out=gpurun_out/r5x; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for rep in 1 2 3 4; do
bash tools/sweep_variants.sh "--no-c4 --no-live-pmc --no-sorted" default tools/_variants/dma_nt.so tools/_variants/dma_aux3.so 2>&1 | grep "q/s" | sed "s/^/human rep$rep /"
done
