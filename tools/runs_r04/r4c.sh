#!/bin/bash
# round 4: (1) parity with the side array + XCD-span dealing, (2) the repeat-bearing C4 workload with / without the side array,
# (3) how coarse may a bucket pass be now that an XCD works through one span of the batch?
out=gpurun_out/r4c; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -5 $out/parity.log
[ $rc -eq 0 ] || echo "PARITY FAILED (continuing with the measurements)"
for side in on off; do
  extra=""; [ $side = off ] && extra="--no-table-side"
  timeout -k 10 600 python bench.py --workload c4r --steps 10 --warmup 2 --counters --no-cpu-baseline --stats-sample 200000 $extra > $out/c4r_side_$side.json 2> $out/c4r_side_$side.err || exit 1
  echo "c4r side=$side $(python -c "import json;d=json.load(open('$out/c4r_side_$side.json'));print(d['value'], d['roofline']['kernel_ms'], d['parity'], json.dumps(d['search_counters']))")"
done
for B in -1 8 12 16 20 24 0; do
  if [ $B -lt 0 ]; then extra=""; else extra="--sort-queries --sort-bits $B"; fi
  timeout -k 10 300 python bench.py --workload c4 --query-kind reads --no-oracle --steps 10 --warmup 2 $extra > $out/c4_B$B.json 2> $out/c4_B$B.err || exit 1
  echo "c4 B=$B $(python -c "import json;d=json.load(open('$out/c4_B$B.json'));print(d['value'], d['roofline']['kernel_ms'])")"
done
timeout -k 10 400 python bench.py --no-oracle --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 > $out/human.json 2> $out/human.err || exit 1
echo "human $(python -c "import json;d=json.load(open('$out/human.json'));print(d['value'], d['roofline']['kernel_ms'])")"
