#!/bin/bash
# round 4: rocprofv3 evidence on the round's final sources: the default line and the repeat-bearing C4 line
O=gpurun_out/r4m; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || exit 1
tools/profile_bench.sh r04_v1 human 2> $O/prof_human.err; tail -2 $O/prof_human.err
tools/profile_bench.sh r04_v1 c4r_reads --workload c4r 2> $O/prof_c4r.err; tail -2 $O/prof_c4r.err
