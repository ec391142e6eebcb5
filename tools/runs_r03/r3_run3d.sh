#!/bin/bash
# round-3 GPU call 3d: parity suite on the final sources; clean kernel stats for the human line; the RCCL code paths on one rank
# at full size; a 5.8e9-symbol real MSBWT (three times C4) with its counters
set -o pipefail
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
PROF_PASSES=stats tools/profile_bench.sh r03_v3 human 2> $O/prof_human_stats.err; tail -1 $O/prof_human_stats.err
echo "== force-dist at full size (1-rank RCCL: torch all_gather, native all-gather, sharded c5)" &&
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --force-dist --no-oracle --steps 10 2> $O/force_dist.err | tee $O/force_dist.json | cut -c1-300; grep -E "native|Error|error" $O/force_dist.err | tail -5
echo "== c4x3: 5.84e9-symbol real MSBWT" && SYNTH_VERBOSE=1 python bench.py --workload c4x3 --steps 10 --cpu-sample 200000 2> $O/c4x3.err | tee $O/c4x3.json | cut -c1-600; grep -E "synth|index file|symbols on the GPU|oracle" $O/c4x3.err | tail -12
PROF_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum" tools/profile_bench.sh r03_v3 c4x3_reads --workload c4x3 2> $O/prof_c4x3.err; tail -1 $O/prof_c4x3.err
