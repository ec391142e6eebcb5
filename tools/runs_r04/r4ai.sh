#!/bin/bash
# round 4, final sources (lanes kernel reworked): rocprofv3 evidence -- the default line, the repeat-bearing C4 line, the random-genome C4 line
O=gpurun_out/r4ai; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || exit 1
tools/profile_bench.sh r04_v4 human 2> $O/prof_human.err; tail -1 $O/prof_human.err
tools/profile_bench.sh r04_v4 c4r_reads --workload c4r 2> $O/prof_c4r.err; tail -1 $O/prof_c4r.err
PROF_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum" tools/profile_bench.sh r04_v4 c4_reads --workload c4 --query-kind reads 2> $O/prof_c4.err; tail -1 $O/prof_c4.err
