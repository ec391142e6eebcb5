#!/bin/bash
# round 4, final sources (lanes kernel reworked): fresh PMC summaries for the other configs (C2, C3 matrix, C3 fused)
O=gpurun_out/r4aj; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || exit 1
tools/profile_bench.sh r04_v4 c2 --workload c2 2> $O/prof_c2.err; tail -1 $O/prof_c2.err
tools/profile_bench.sh r04_v4 c3 --workload c3 2> $O/prof_c3.err; tail -1 $O/prof_c3.err
tools/profile_bench.sh r04_v4 c3_fused --workload c3 --fused 2> $O/prof_c3f.err; tail -1 $O/prof_c3f.err
