#!/bin/bash
# round-3 GPU call 7: SWAR query packing: parity suite, then the lines it should move
set -o pipefail
O=gpurun_out/r3k; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 || tail -60 $O/pytest.log
tail -3 $O/pytest.log
echo "== C2" && python bench.py --workload c2 --no-oracle 2> $O/c2.err | tee $O/c2.json | cut -c1-200
echo "== C2, 5e7 random 21-mers" && python bench.py --workload c2 --queries 50000000 --no-oracle 2> $O/c2b.err | tee $O/c2b.json | cut -c1-200
echo "== C3 fused" && python bench.py --workload c3 --fused --no-oracle 2> $O/c3f.err | tee $O/c3f.json | cut -c1-200
echo "== C4 random" && python bench.py --workload c4 --no-oracle 2> $O/c4.err | tee $O/c4.json | cut -c1-200
echo "== human + c5" && python bench.py --no-oracle --no-c4 --steps 10 2> $O/human.err | tee $O/human.json | cut -c1-200; python -c "import json; r=json.loads(open('$O/human.json').read()); print(r['c5_random_1e9'])"
