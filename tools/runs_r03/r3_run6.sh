#!/bin/bash
# round-3 GPU call 6: smoke(), then a complete default run including the live PMC passes
set -o pipefail
O=gpurun_out/r3j; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
T0=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: rc=$?, $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r3j/bench_default.json").read())
ro=r["roofline"]
print("value %.4e frac %.4f traffic_source %s" % (r["value"], ro["frac"], ro["traffic_source"]))
print(ro.get("traffic_live"))
PY
grep -E "live PMC|c4 line:|exact MSBWT" $O/bench_default.err | cut -c1-200
