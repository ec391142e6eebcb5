#!/bin/bash
# round-3 GPU call 16: final validation of the committed tree -- the whole suite, smoke(), a complete default run
set -o pipefail
O=gpurun_out/r3u; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: rc=$?, $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r3u/bench_default.json").read())
print("value %.4e frac %.4f parity %s"%(r["value"], r["roofline"]["frac"], r["parity"])); print(r["roofline"]["traffic_live"]); print(r["sorted_batch"]["value"], r["c5_random_1e9"]["value"], r["c4_real_reads"]["value"], r["c4_real_reads"]["roofline"]["frac"])
PY
