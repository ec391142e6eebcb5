#!/bin/bash
# round-3 GPU call 15: experiment -- ALL windows of C3's reads as a query matrix, as generated (read order) and ordered by key:
# the ceiling of an ordered traversal for the fused path
set -o pipefail
O=gpurun_out/r3t; mkdir -p $O
echo "== all windows, matrix, read order" && python bench.py --workload c3 --queries 185666040 --no-oracle --steps 10 2> $O/a.err | tee $O/c3_allwin_plain.json | cut -c1-170
echo "== all windows, matrix, ordered" && python bench.py --workload c3 --queries 185666040 --no-oracle --steps 10 --sort-queries 2> $O/b.err | tee $O/c3_allwin_sorted.json | cut -c1-170
echo "== fused" && python bench.py --workload c3 --fused --no-oracle --steps 10 2> $O/c.err | tee $O/c3_fused.json | cut -c1-170
