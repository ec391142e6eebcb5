#!/bin/bash
out=gpurun_out/r5u; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py tests/test_cpp_mirror.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $out/tests.log
