#!/bin/bash
# round-3 GPU call 13: complete default run after the live-PMC child was told to skip the ordered-batch extra
set -o pipefail
O=gpurun_out/r3r; mkdir -p $O
T0=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: rc=$?, $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r3r/bench_default.json").read())
print("value %.4e frac %.4f"%(r["value"], r["roofline"]["frac"])); print(r["roofline"]["traffic_live"]); print(r["sorted_batch"]["value"], r["c5_random_1e9"]["value"], r["c4_real_reads"]["value"])
PY
