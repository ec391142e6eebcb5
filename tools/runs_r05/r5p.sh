#!/bin/bash
# round 5: the pipelined count + all-gather on real RCCL (one rank), the bench's N > 1 path rehearsed with --force-dist at full size
out=gpurun_out/r5p; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bench_smoke.py -q -m gpu -k "pipelined or allgather or rccl or bench" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" $out/tests.log | tail -8
[ $rc -eq 0 ] || tail -30 $out/tests.log
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 600 python bench.py --force-dist --no-c5 --no-c4 --no-live-pmc --no-sorted --no-cpu-baseline --steps 10 --warmup 2 --parity-sample 200000 --stats-sample 200000 > $out/force_dist.json 2> $out/force_dist.err || { tail -5 $out/force_dist.err; exit 1; }
python - <<PY
import json
d=json.load(open("$out/force_dist.json"))
print("value", d["value"], "ms", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"])
print(json.dumps({k:v for k,v in d["native_gather"].items() if "note" not in k}))
print(json.dumps(d.get("ranks")))
PY
