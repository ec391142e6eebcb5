#!/bin/bash
# round 5: the new GPU tests (run-block fallback, per-thread streams, memory budget on a loaded index), the sparse tests after the re-sizing,
# then the default bench run as the driver makes it (timed)
out=gpurun_out/r5k; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py -q -m gpu -k "sparse or fall_back or per_thread or memory_budget or run_block" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" $out/tests.log | tail -15
t0=$(date +%s)
MSBWT_VERBOSE=1 timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err; rc=$?
t1=$(date +%s); echo "default run rc=$rc in $((t1-t0)) s"
grep -E "sparse table:|live PMC|c4_|PARITY" $out/bench_default.err | tail -20
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print("value %.4g q/s, %.2f ms/step, kernel %.3f ms, frac %s" % (d["value"] or -1, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
print("layout", {k:v for k,v in d["roofline"]["layout_algorithmic"].items() if k!="note"})
print("parity", d["parity"])
for k in ("sorted_batch","c5_random_1e9","c4_repeats","c4_real_reads"):
    v=d.get(k); print(k, None if v is None else {kk:v[kk] for kk in ("value","ms_per_step") if kk in v}, (v or {}).get("parity",{}).get("mismatches"), (v or {}).get("library_ordered",{}).get("value"), ((v or {}).get("roofline") or {}).get("frac"), ((v or {}).get("pair_blocks_rebuilt") or {}).get("ratio_to_the_line"))
cb=(d.get("c4_repeats") or {}).get("copy_number_bins")
if cb:
    for b in cb["bins"]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in b.items()})
PY
