#!/bin/bash
# round 4: randomised soak over the final sources (side array, packed queries, library order, budgets, run blocks on the lanes kernel), then a C4 probe with telemetry
out=gpurun_out/r4s; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for seed in 51 52; do
  STRESS_SEED=$seed timeout -k 10 420 python tools/stress_parity.py 330 > $out/soak_seed$seed.log 2>&1; echo "soak seed $seed rc=$?"; tail -2 $out/soak_seed$seed.log
done
timeout -k 10 300 python bench.py --workload c4 --query-kind reads --no-oracle --steps 10 --warmup 2 > $out/c4_probe.json 2> $out/c4_probe.err
python -c "import json;d=json.load(open('$out/c4_probe.json'));print('c4 probe %.4g q/s %.2f ms' % (d['value'], d['ms_per_step']), json.dumps(d['telemetry']['during_timed_region']))"
