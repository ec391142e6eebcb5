#!/bin/bash
# round-3 GPU call 14: the bench smoke tests (multi-rank rehearsals included) after the consistency checks became reported flags
set -o pipefail
O=gpurun_out/r3s; mkdir -p $O
python -m pytest tests/test_bench_smoke.py -m gpu -q 2>&1 | tail -4
MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --force-dist --no-oracle --steps 5 2> $O/fd.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r.get('consistency_errors'), r['native_gather'].get('value'), r['ranks']['exchange_ms_alone'])"
