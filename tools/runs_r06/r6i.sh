#!/bin/bash
# round 6: msbwt_rle_set_sparse_second -- its test, the C++ mirror (new methods), the shim check on the box
out=gpurun_out/r6i; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout -k 10 500 python -m pytest tests/test_gpu_sparse.py tests/test_cpp_mirror.py -x -q -m gpu -k "second or mirror or automatic" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || { grep -n "^E  \|FAILED" $out/tests.log | head -30; exit $rc; }
