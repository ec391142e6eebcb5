#!/bin/bash
# round 5, final sources: rocprofv3 evidence for the default line (human scale) and for the run-block format at human scale (profiles/r05_v3)
bash tools/profile_bench.sh r05_v3 human 2>&1 | tail -2
bash tools/profile_bench.sh r05_v3 human_runs --blocks runs --queries 100000000 2>&1 | tail -2
