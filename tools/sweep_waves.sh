#!/bin/bash
# throughput of the lanes kernel against resident waves per CU
#   usage: tools/sweep_waves.sh "<bench.py arguments>" <waves/CU>...
ARGS=$1; shift
for w in "$@"; do
  MSBWT_VERBOSE=1 MSBWT_LANES_WAVES_PER_CU=$w timeout -k 10 200 python3 bench.py $ARGS --no-oracle --no-c5 --steps 10 --warmup 2 2> /tmp/sweep.err | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('waves/CU $w', 'q/s %.4e' % r['value'], 'ms', round(r['ms_per_step'],2))"
  grep "msbwt\]" /tmp/sweep.err | sort -u
done
