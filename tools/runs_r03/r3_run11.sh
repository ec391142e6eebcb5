#!/bin/bash
# round-3 GPU call 11: suite + complete default run with the sorted_batch key
set -o pipefail
O=gpurun_out/r3p; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
T0=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: rc=$?, $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r3p/bench_default.json").read())
print("value %.4e frac %.4f"%(r["value"], r["roofline"]["frac"])); print(r["sorted_batch"]); print(r["c5_random_1e9"]["value"], r["c4_real_reads"]["value"])
PY
