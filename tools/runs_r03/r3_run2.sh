#!/bin/bash
# round-3 GPU call 2: parity suite on the split-straddle kernel + new bench.py, A/B of the kernel variants, launch floor
set -o pipefail
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
tools/ubench_launch > $O/launch.log 2>&1; cat $O/launch.log
for W in 12 16; do
  echo "== split, $W waves: human" && MSBWT_LANES_WAVES_PER_CU=$W python bench.py --no-oracle --no-c5 --no-c4 --steps 10 2> $O/human_$W.err | tee $O/human_$W.json | cut -c1-200 &&
  echo "== split, $W waves: c3 fused" && MSBWT_LANES_WAVES_PER_CU=$W python bench.py --workload c3 --fused --no-oracle 2> $O/c3f_$W.err | tee $O/c3f_$W.json | cut -c1-200 &&
  echo "== split, $W waves: c4 reads" && MSBWT_LANES_WAVES_PER_CU=$W python bench.py --workload c4 --query-kind reads --no-oracle 2> $O/c4r_$W.err | tee $O/c4r_$W.json | cut -c1-200 || exit 1
done
echo "== split, k=59" && python bench.py --k 59 --queries 100000000 --no-oracle --no-c5 --no-c4 --steps 10 2> $O/k59.err | tee $O/k59.json | cut -c1-200
cp rust-msbwt_amd/libmsbwt_hip.so /tmp/lib_default.so && cp tools/_variants/nosplit.so rust-msbwt_amd/libmsbwt_hip.so &&
echo "== nosplit: human" && python bench.py --no-oracle --no-c5 --no-c4 --steps 10 2> $O/human_ns.err | tee $O/human_ns.json | cut -c1-200 &&
echo "== nosplit: c3 fused" && python bench.py --workload c3 --fused --no-oracle 2> $O/c3f_ns.err | tee $O/c3f_ns.json | cut -c1-200 &&
echo "== nosplit: c4 reads" && python bench.py --workload c4 --query-kind reads --no-oracle 2> $O/c4r_ns.err | tee $O/c4r_ns.json | cut -c1-200
cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so
grep -h "load:" $O/human_12.err | tail -6
