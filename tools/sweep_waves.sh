#!/bin/bash
# throughput of the lanes kernel against resident waves per CU (human-scale index, present 31-mers)
for w in "$@"; do
  MSBWT_VERBOSE=1 MSBWT_LANES_WAVES_PER_CU=$w timeout -k 10 200 python3 bench.py --no-oracle --no-c5 --queries 100000000 --steps 10 --warmup 2 2> /tmp/sweep.err | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('waves/CU $w', 'q/s %.4e' % r['value'], 'ms', round(r['ms_per_step'],2))"
  grep "msbwt\]" /tmp/sweep.err | sort -u
done
