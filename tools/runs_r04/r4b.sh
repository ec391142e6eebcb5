#!/bin/bash
out=gpurun_out/r4b; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 300 python tools/runs_r04/scatter_probe.py > $out/scatter_probe.log 2>&1 || exit 1
cat $out/scatter_probe.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/parity.log 2>&1; echo "parity rc=$?"; tail -5 $out/parity.log
