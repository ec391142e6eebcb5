#!/bin/bash
# round-3 GPU call 3c: the full default bench run on the exact-BWT human-scale workload, then the rocprofv3 evidence for it and for the C4 line
set -o pipefail
O=gpurun_out/r3e; mkdir -p $O
T0=$(date +%s)
MSBWT_VERBOSE=1 python bench.py > $O/bench_default.json 2> $O/bench_default.err; RC=$?
echo "default bench: rc=$RC, $(( $(date +%s) - T0 )) s"; cut -c1-1500 $O/bench_default.json; grep -E "exact MSBWT|load:|symbols on the GPU|c4 line|c5 line|oracle loaded|Error|error" $O/bench_default.err | tail -24
[ $RC -eq 0 ] || exit 1
tools/profile_bench.sh r03_v3 human 2> $O/prof_human.err && tail -2 $O/prof_human.err &&
tools/profile_bench.sh r03_v3 c4_reads --workload c4 --query-kind reads 2> $O/prof_c4.err && tail -2 $O/prof_c4.err
