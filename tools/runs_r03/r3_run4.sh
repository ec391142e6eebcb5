#!/bin/bash
# round-3 GPU call 4: the whole suite on the final sources, remaining measurements, fresh profiles for C2 / C3, a full default run
set -o pipefail
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
NPY=$(ls synth/cache/c2_*_comp_msbwt.npy | head -1)
echo "== call latency" && gcc -O2 -Iinclude examples/call_latency.c -Lrust-msbwt_amd -lmsbwt_hip -Wl,-rpath,$PWD/rust-msbwt_amd -o /tmp/cl_new && /tmp/cl_new $NPY 21 | tee $O/latency_r03.log | head -4
echo "== C3 fused, table depth 17 (MSBWT_TABLE_DEPTH=15 MSBWT_TABLE_PACKED=1)" && MSBWT_TABLE_DEPTH=15 MSBWT_TABLE_PACKED=1 python bench.py --workload c3 --fused --no-oracle 2> $O/c3f_d17.err | tee $O/c3f_d17.json | cut -c1-200
echo "== C3 fused, default" && python bench.py --workload c3 --fused --no-oracle 2> $O/c3f.err | tee $O/c3f.json | cut -c1-200
echo "== C2 default" && python bench.py --workload c2 2> $O/c2.err | tee $O/c2.json | cut -c1-200
tools/profile_bench.sh r03_v3 c2 --workload c2 2> $O/prof_c2.err; tail -1 $O/prof_c2.err
tools/profile_bench.sh r03_v3 c3 --workload c3 2> $O/prof_c3.err; tail -1 $O/prof_c3.err
T0=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: rc=$?, $(( $(date +%s) - T0 )) s"; cut -c1-400 $O/bench_default.json
