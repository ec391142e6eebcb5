#!/bin/bash
# round-3 GPU call 3a: parity suite on the final kernels; before/after (round-2 tree) for k = 59 and call latency; C3 fused profile
set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
NPY=$(ls synth/cache/c2_*_comp_msbwt.npy | head -1)
echo "== call latency, round 3" && gcc -O2 -Iinclude examples/call_latency.c -Lrust-msbwt_amd -lmsbwt_hip -Wl,-rpath,$PWD/rust-msbwt_amd -o /tmp/cl_new && /tmp/cl_new $NPY 21 | tee $O/latency_r03.log
echo "== call latency, round 2 tree" && gcc -O2 -I.r02_tree/include examples/call_latency.c -L.r02_tree/rust-msbwt_amd -lmsbwt_hip -Wl,-rpath,$PWD/.r02_tree/rust-msbwt_amd -o /tmp/cl_old && /tmp/cl_old $NPY 21 | tee $O/latency_r02.log
echo "== k=59 human, round 3 (geometric stream, as round 2)" && python bench.py --k 59 --queries 100000000 --stream geometric --no-oracle --no-c5 --no-c4 --steps 10 2> $O/k59_r03.err | tee $O/k59_r03.json | cut -c1-220
echo "== k=59 human, round 2 tree" && (cd .r02_tree && python bench.py --k 59 --queries 100000000 --no-oracle --no-c5 --steps 10 2> ../$O/k59_r02.err | tee ../$O/k59_r02.json | cut -c1-220)
echo "== host API" && python tools/host_api_bench.py 20000000 > $O/host_api.log 2>&1; grep -v "^msbwt_rle" $O/host_api.log | tail -12
tools/profile_bench.sh r03_v2 c3_fused --workload c3 --fused 2> $O/prof_c3f.err; tail -2 $O/prof_c3f.err
