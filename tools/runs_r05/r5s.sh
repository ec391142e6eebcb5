#!/bin/bash
# round 5, kernel revision 2: rocprofv3 evidence for C2 and C3 fused (profiles/r05_v3)
bash tools/profile_bench.sh r05_v3 c2 --workload c2 2>&1 | tail -2
bash tools/profile_bench.sh r05_v3 c3_fused --workload c3 --fused 2>&1 | tail -2
