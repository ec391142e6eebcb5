#!/bin/bash
# round-3 GPU call 1: parity suite, C4 run histogram, quick speed checks of the reworked kernels
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/measure_run_histogram.py c4 $O/c4_run_histogram.json > $O/hist.log 2>&1 && tail -1 $O/hist.log &&
MSBWT_VERBOSE=1 python bench.py --no-oracle --no-c5 --steps 10 > $O/human.json 2> $O/human.err && cat $O/human.json &&
python bench.py --workload c3 --fused --no-oracle > $O/c3f.json 2> $O/c3f.err && cat $O/c3f.json &&
PROF_PASSES=SQ_WAVE_CYCLES tools/profile_bench.sh r03_v1 c3_fused --workload c3 --fused 2> $O/prof.err &&
python bench.py --k 59 --queries 100000000 --no-oracle --no-c5 --steps 10 > $O/human_k59.json 2> $O/human_k59.err && cat $O/human_k59.json &&
python tools/host_api_bench.py 20000000 > $O/host_api.log 2>&1; tail -25 $O/host_api.log
