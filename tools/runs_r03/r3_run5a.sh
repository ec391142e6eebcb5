#!/bin/bash
# round-3 GPU call 5a: rocprofv3 evidence on the FINAL sources: human (exact BWT), C4 read-derived, C3 fused
set -o pipefail
O=gpurun_out/r3l; mkdir -p $O
tools/profile_bench.sh r03_v5 human 2> $O/prof_human.err; tail -1 $O/prof_human.err
tools/profile_bench.sh r03_v5 c4_reads --workload c4 --query-kind reads 2> $O/prof_c4.err; tail -1 $O/prof_c4.err
tools/profile_bench.sh r03_v5 c3_fused --workload c3 --fused 2> $O/prof_c3f.err; tail -1 $O/prof_c3f.err
