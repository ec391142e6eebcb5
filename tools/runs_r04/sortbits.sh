#!/bin/bash
# round 4 experiment: how many key bits does a bucket pass need?  (batch ordered OUTSIDE the timed region by the top B bits)
out=gpurun_out/r4a; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
for B in -1 8 12 16 20 24 28 34 0; do
  if [ $B -lt 0 ]; then extra=""; else extra="--sort-queries --sort-bits $B"; fi
  timeout -k 10 300 python bench.py --workload c4 --query-kind reads --no-oracle --steps 10 --warmup 2 $extra > $out/c4_B$B.json 2> $out/c4_B$B.err || exit 1
  echo "c4 B=$B $(python -c "import json;d=json.load(open('$out/c4_B$B.json'));print(d['value'], d['roofline']['kernel_ms'])")"
done
for B in -1 12 16 20 24 0; do
  if [ $B -lt 0 ]; then extra=""; else extra="--sort-queries --sort-bits $B"; fi
  timeout -k 10 400 python bench.py --no-oracle --no-c5 --no-c4 --no-live-pmc --no-sorted --steps 10 --warmup 2 $extra > $out/human_B$B.json 2> $out/human_B$B.err || exit 1
  echo "human B=$B $(python -c "import json;d=json.load(open('$out/human_B$B.json'));print(d['value'], d['roofline']['kernel_ms'])")"
done
