#!/bin/bash
# throughput of prebuilt library variants (tools/_variants/*.so, built with other -DMSBWT_LANES_* values)
#   usage: tools/sweep_variants.sh "<bench.py arguments>" <variant.so>...     ("default" = the in-tree build)
ARGS=$1; shift
cp rust-msbwt_amd/libmsbwt_hip.so /tmp/lib_default.so
for v in "$@"; do
  if [ "$v" = default ]; then cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so; else cp "$v" rust-msbwt_amd/libmsbwt_hip.so; fi
  MSBWT_VERBOSE=1 timeout -k 10 400 python3 bench.py $ARGS --no-oracle --no-c5 --no-cpu-baseline --steps 10 --warmup 2 2> /tmp/sweep.err | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'q/s %.4e' % r['value'], 'ms', round(r['ms_per_step'],2))" || tail -5 /tmp/sweep.err
  grep "lanes kernel" /tmp/sweep.err | sort -u
done
cp /tmp/lib_default.so rust-msbwt_amd/libmsbwt_hip.so
