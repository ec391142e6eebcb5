#!/bin/bash
# round 5, final sources: rocprofv3 evidence for the two C4-sized lines (profiles/r05_v3)
bash tools/profile_bench.sh r05_v3 c4_reads --workload c4 --query-kind reads 2>&1 | tail -2
bash tools/profile_bench.sh r05_v3 c4r_reads --workload c4r 2>&1 | tail -2
