#!/bin/bash
# round 5: final validation once more after the bench.py refactor (copy_number_bins helper, --genome) and the capi.cpp clean-ups: GPU suite, smoke(), the default bench command
out=gpurun_out/r5final2; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $out/gputests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -4 $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
t0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err; rc=$?
t1=$(date +%s); echo "default run rc=$rc in $((t1-t0)) s"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print("value %.4g q/s, %.2f ms/step, kernel %.3f ms, frac %.3f, traffic_over_algorithmic %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["layout_algorithmic"]["traffic_over_algorithmic"]))
print("parity", d["parity"], "cpu", d["cpu_baseline"]["value"])
for k in ("sorted_batch","c5_random_1e9","c4_repeats","c4_real_reads"):
    v=d.get(k); print(k, None if v is None else {kk:v[kk] for kk in ("value","ms_per_step") if kk in v}, (v or {}).get("parity",{}).get("mismatches"), (v or {}).get("library_ordered",{}).get("value"), ((v or {}).get("roofline") or {}).get("frac"), (((v or {}).get("roofline") or {}).get("random_lines") or {}).get("frac"), ((v or {}).get("pair_blocks_rebuilt") or {}).get("ratio_to_the_line"))
PY
