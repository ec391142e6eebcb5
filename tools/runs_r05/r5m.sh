#!/bin/bash
# round 5, final sources: the whole GPU suite, then rocprofv3 evidence for C2 and C3 fused (profiles/r05_v1)
out=gpurun_out/r5m; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 700 python -m pytest tests -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" $out/gputests.log | tail -10
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
bash tools/profile_bench.sh r05_v1 c2 --workload c2 2>&1 | tail -2
bash tools/profile_bench.sh r05_v1 c3_fused --workload c3 --fused 2>&1 | tail -2
