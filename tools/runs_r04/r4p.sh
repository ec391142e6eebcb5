#!/bin/bash
# run-block decode variants on one box: human-scale run blocks, 1e8 present 31-mers
out=gpurun_out/r4p; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
MSBWT_SEARCH=lanes tools/sweep_variants.sh "--blocks runs --no-c4 --no-live-pmc --no-sorted --queries 100000000" default tools/_variants/sep6.so tools/_variants/sep3.so tools/_variants/fused2.so tools/_variants/fused1.so 2>&1 | grep -E "q/s|error" | tee $out/variants.log
