#!/bin/bash
# round-3 GPU call 10: experiment -- batches ordered by table index
set -o pipefail
O=gpurun_out/r3o; mkdir -p $O
for W in "c4 --query-kind reads" "c3"; do
  set -- $W
  echo "== $W, as generated" && python bench.py --workload $W --no-oracle 2> $O/a.err | tee $O/${1}_plain.json | cut -c1-170
  echo "== $W, sorted" && python bench.py --workload $W --no-oracle --sort-queries 2> $O/b.err | tee $O/${1}_sorted.json | cut -c1-170
done
echo "== human, sorted" && python bench.py --no-oracle --no-c5 --no-c4 --steps 10 --sort-queries 2> $O/h.err | tee $O/human_sorted.json | cut -c1-170
echo "== human, as generated" && python bench.py --no-oracle --no-c5 --no-c4 --steps 10 2> $O/h2.err | tee $O/human_plain.json | cut -c1-170
