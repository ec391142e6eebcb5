#!/bin/bash
# round 6: the two-tier sparse table and the sparse table behind run blocks, first time on a GPU -- their own tests, then the C4-sized one, then bench smoke
out=gpurun_out/r6b; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 500 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu ${SPARSE_K:+-k "$SPARSE_K"} > $out/sparse.log 2>&1; rc=$?; echo "sparse tests rc=$rc"; tail -3 $out/sparse.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert \|FAILED\|^E " $out/sparse.log | head -40; exit $rc; }
timeout -k 10 400 python -m pytest tests/test_gpu_config_sizes.py -x -q -m gpu -k "c4_real" > $out/c4.log 2>&1; rc=$?; echo "c4 test rc=$rc"; tail -3 $out/c4.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert \|FAILED\|^E " $out/c4.log | head -40; exit $rc; }
timeout -k 10 500 python -m pytest tests/test_bench_smoke.py -x -q -m gpu > $out/smoke.log 2>&1; rc=$?; echo "bench smoke rc=$rc"; tail -3 $out/smoke.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert \|FAILED\|^E " $out/smoke.log | head -40; exit $rc; }
