#!/bin/bash
# round-3 GPU call 5b: the suite and a complete default run on the FINAL sources; fresh evidence for C2, C3 (matrix queries) and c4x3
set -o pipefail
O=gpurun_out/r3m; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
T0=$(date +%s); MSBWT_VERBOSE=1 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: rc=$?, $(( $(date +%s) - T0 )) s"; cut -c1-300 $O/bench_default.json
tools/profile_bench.sh r03_v5 c2 --workload c2 2> $O/prof_c2.err; tail -1 $O/prof_c2.err
tools/profile_bench.sh r03_v5 c3 --workload c3 2> $O/prof_c3.err; tail -1 $O/prof_c3.err
PROF_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum" tools/profile_bench.sh r03_v5 c4x3_reads --workload c4x3 2> $O/prof_c4x3.err; tail -1 $O/prof_c4x3.err
