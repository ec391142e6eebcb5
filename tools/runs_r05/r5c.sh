#!/bin/bash
# round 5: the whole GPU suite (no -x: every failure at once)
out=gpurun_out/r5c; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 1100 python -m pytest tests -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" $out/gputests.log | tail -40
