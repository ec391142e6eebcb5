#!/bin/bash
# the driver's own command on the final sources: the whole default run, timed
out=gpurun_out/r4r; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
t0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err; rc=$?
t1=$(date +%s); echo "default run rc=$rc in $((t1-t0)) s"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print("value %.4g q/s, %.2f ms/step, kernel %.3f ms, frac %.3f, traffic_over_algorithmic %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["layout_algorithmic"]["traffic_over_algorithmic"]))
print("parity", d["parity"], "cpu", d["cpu_baseline"]["value"])
for k in ("sorted_batch","c5_random_1e9","c4_repeats","c4_real_reads"):
    v=d.get(k); print(k, None if v is None else {kk:v[kk] for kk in ("value","ms_per_step","parity") if kk in v}, (v or {}).get("library_ordered",{}).get("value"))
print("c4_repeats counters", {k:round(v,4) if isinstance(v,float) else v for k,v in d["c4_repeats"]["search_counters"].items() if k not in ("raw","note","table")})
print("telemetry during", d["telemetry"]["during_timed_region"])
print("c4r telemetry", d["c4_repeats"]["telemetry"]["during_timed_region"])
PY
grep -E "^\[bench\]" $out/bench_default.err | cut -c1-200 | tail -40
