#!/bin/bash
# round 6: one kernel per sparse-table layout (lanes_kernel.hpp in four translation units): the sparse tests, then the same-box A/B against round 5's tree again
out=gpurun_out/r6g; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 500 python -m pytest tests/test_gpu_sparse.py tests/test_cpp_mirror.py -x -q -m gpu ${SPARSE_K:+-k "$SPARSE_K"} > $out/sparse.log 2>&1; rc=$?; echo "sparse tests rc=$rc"; tail -2 $out/sparse.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/sparse.log | head -30; exit $rc; }
bash tools/runs_r06/r6c.sh
