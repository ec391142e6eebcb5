#!/bin/bash
# round-3 GPU call 3b: the exact-BWT human-scale workload: first at a tenth of the size with full parity, then at full size
set -o pipefail
O=gpurun_out/r3d; mkdir -p $O
echo "== exact BWT, scale 0.1, with oracle" && python bench.py --scale 0.1 --queries 50000000 --no-c5 --no-c4 --steps 5 --cpu-sample 200000 2> $O/bwt_01.err | tee $O/bwt_01.json | cut -c1-400; grep -E "bwt:|exact MSBWT|load:|symbols on the GPU|oracle" $O/bwt_01.err | tail -20
echo "== exact BWT, full scale" && MSBWT_VERBOSE=1 python bench.py --no-oracle --no-c5 --no-c4 --steps 10 2> $O/bwt_full.err | tee $O/bwt_full.json | cut -c1-600; grep -E "bwt:|exact MSBWT|load:|symbols on the GPU|Error|error" $O/bwt_full.err | tail -30
echo "== histogram stream, full scale" && python bench.py --stream histogram --no-oracle --no-c5 --no-c4 --steps 10 2> $O/hist_full.err | tee $O/hist_full.json | cut -c1-300
echo "== host API, round-2 tree" && (cd .r02_tree && python tools/host_api_bench.py 20000000 2>&1 | grep "host API" | tail -4)
