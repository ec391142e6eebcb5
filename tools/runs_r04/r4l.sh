#!/bin/bash
# round 4: the whole GPU suite on the sources as they are, then the host-API and memory-budget measurements
out=gpurun_out/r4l; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -8 $out/gputests.log
timeout -k 10 600 python tools/host_api_bench.py > $out/host_api.log 2>&1; echo "host api rc=$?"; grep -E "host API|parity|us per" $out/host_api.log
for w in c2 c4; do timeout -k 10 400 python tools/budget_bench.py $w > $out/budget_$w.log 2>&1; echo "budget $w rc=$?"; cat $out/budget_$w.log | cut -c1-400; done
